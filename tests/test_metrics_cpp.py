"""host/eppk_metrics.hpp (the extractor half of the metrics scraper: Prometheus text of a model server -> eppk_pod_row, after
docs/proposals/003-model-server-protocol/README.md) compiled with g++ and run on the CPU: tests/cpp/test_metrics.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_metrics.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_metrics")


def test_metrics_extractor():
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", SRC, "-o", EXE], check=True)
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "metrics ok" in out.stdout
