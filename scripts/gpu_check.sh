#!/bin/bash
# Run on the GPU box (via gpurun): smoke + GPU parity tests.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
rocminfo | grep -m2 gfx
echo "nproc=$(nproc)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
