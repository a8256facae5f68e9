#!/bin/bash
# Round-6 campaigns (GPU box): the randomized tests over seeds beyond the suite's, in the library modes, on the protocol-v5 sources.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r6_campaigns; mkdir -p $OUT
{
echo "== fuzz campaign (default mode)";            timeout 400 python scripts/gpu_fuzz_campaign.py 120 600000 1600000 2>&1 | tail -2
echo "== fuzz campaign, EPPK_QUAD_MIN=4";          EPPK_QUAD_MIN=4 timeout 400 python scripts/gpu_fuzz_campaign.py 100 610000 1610000 2>&1 | tail -2
echo "== fuzz campaign, EPPK_LISTS=0";             EPPK_LISTS=0 timeout 300 python scripts/gpu_fuzz_campaign.py 40 620000 1620000 2>&1 | tail -2
echo "== fuzz campaign, EPPK_QUAD=0";              EPPK_QUAD=0 timeout 300 python scripts/gpu_fuzz_campaign.py 40 630000 1630000 2>&1 | tail -2
echo "== stage campaign";                          timeout 300 python scripts/gpu_stage_campaign.py 60 7000 2>&1 | tail -2
echo "== stage campaign, EPPK_QUAD_MIN=4";         EPPK_QUAD_MIN=4 timeout 300 python scripts/gpu_stage_campaign.py 60 8000 2>&1 | tail -2
echo "== long closed loop, 16k requests";          timeout 300 python scripts/gpu_closed_loop_long.py 100 16384 2>&1 | tail -2
echo "== long closed loop, full size";             timeout 600 python scripts/gpu_closed_loop_long.py 120 65536 2>&1 | tail -2
echo "== fuzz campaign under EPPK_RESIDENT=1";     EPPK_RESIDENT=1 timeout 300 python scripts/gpu_fuzz_campaign.py 60 640000 1640000 2>&1 | tail -2
echo "== stage campaign, EPPK_RESIDENT=1";         EPPK_RESIDENT=1 timeout 300 python scripts/gpu_stage_campaign.py 60 9000 2>&1 | tail -2
echo "== GPU suite subset under EPPK_RESIDENT=1";  EPPK_RESIDENT=1 timeout 900 python -m pytest tests -m gpu -q --ignore=tests/test_gpu_quad.py -k "not group and not fullsize and not closed_loop and not without_the_switch" 2>&1 | tail -3   # (test_gpu_quad counts LAUNCHES of 5-request batches; the deselected test asserts that nothing is resident without the switch)
} 2>&1 | grep -v "amdgpu.ids" | tee $OUT/campaigns.txt
