"""The Go side ships as a patch against the reference tree (integration/): it must apply cleanly to the reference's own files.
(Uncompiled: no Go toolchain in this image.  Needs /root/reference, so it runs in the build container only.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "0001-lwepp-gpu-endpoint-picker.patch")
REF = "/root/reference"
FILES = ["pkg/lwepp/handlers/server.go", "pkg/lwepp/server/options.go", "pkg/lwepp/server/runserver.go", "cmd/lwepp/main.go", "lwepp.Dockerfile"]


def test_patch_names_only_the_picker_seam():
    text = open(PATCH).read()
    touched = sorted({ln.split("\t")[0][len("+++ b/"):] for ln in text.splitlines() if ln.startswith("+++ b/")})
    assert touched == sorted(FILES + ["pkg/lwepp/handlers/gpupicker.go", "pkg/lwepp/handlers/gpupicker_nocgo.go", "pkg/lwepp/handlers/gpusnapshot.go", "pkg/lwepp/handlers/gpusnapshot_test.go"])
    assert "ctx.Done()" in text and "eppk_index_insert" in text and "runtime.LockOSThread()" in text
    # the picker is fed: main.go starts the snapshot producer (scrape -> pod rows -> PublishSnapshot), in both build modes
    assert "go sp.Run(ctx)" in text and text.count("func (p *GPUPicker) PublishSnapshot(") == 2


def test_go_pod_row_matches_the_c_struct():
    """PodRow of gpusnapshot.go is copied over eppk_pod_row byte for byte: same fields, same order, 64 bytes (the Go file asserts the
    size at compile time; this checks the field order against include/eppk.h here, where no Go toolchain exists)."""
    import re
    text = open(PATCH).read()
    go = re.search(r"type PodRow struct \{(.*?)\n\+\}", text, re.S).group(1)
    go_fields = [ln.lstrip("+").split()[0].lower() for ln in go.splitlines() if ln.lstrip("+").strip() and not ln.lstrip("+").strip().startswith("//")]
    hdr = open(os.path.join(ROOT, "include", "eppk.h")).read()
    c = re.search(r"typedef struct eppk_pod_row \{(.*?)\} eppk_pod_row;", hdr, re.S).group(1)
    c_fields = [re.sub(r"\[.*", "", ln.split(";")[0].split()[-1]).replace("_", "") for ln in c.splitlines() if ";" in ln]
    assert go_fields == c_fields == ["queue", "running", "kvutil", "maxlora", "flags", "active", "waiting", "reserved"]


def test_patch_applies_to_the_reference(tmp_path):
    if not os.path.isdir(REF) or shutil.which("patch") is None:
        pytest.skip("needs /root/reference and patch(1)")
    for f in FILES:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    out = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Hunk" not in out.stdout            # no fuzz, no offsets
    src = open(tmp_path / "pkg/lwepp/handlers/server.go").read()
    assert "func NewStreamingServer(datastore Datastore) *StreamingServer" in src and "NewStreamingServerWithPicker" in src
    # the reference's own tests keep compiling against the unchanged constructor
    assert "NewStreamingServer(ds)" in open(os.path.join(REF, "pkg/lwepp/handlers/request_test.go")).read()


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [a for a in out if a.strip()]


def test_every_cgo_call_in_the_patch_matches_the_header():
    """No Go toolchain here, so the compiler cannot check the cgo calls: every `C.eppk_*(...)` call in the patch must name a function
    include/eppk.h declares, with the declared number of arguments; every `C.EPPK_*` constant must be a #define of the header."""
    import re
    text = "\n".join(ln[1:] for ln in open(PATCH).read().splitlines() if ln.startswith("+") and not ln.startswith("+++"))
    hdr = open(os.path.join(ROOT, "include", "eppk.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(eppk_\w+)\s*\(([^;{]*?)\)\s*;", hdr_nc):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    consts = set(re.findall(r"#define\s+(EPPK_\w+)", hdr)) | set(re.findall(r"\b(EPPK_\w+)\s*=", hdr_nc))
    calls = 0
    for m in re.finditer(r"\bC\.(eppk_\w+)\(", text):
        name = m.group(1)
        if name not in protos:      # a type conversion such as C.eppk_pod_row{} / (*C.eppk_req_hdr)(row) is not a call; anything else must be declared
            assert re.search(r"\b%s\b" % name, hdr), f"{name} is not in include/eppk.h"
            continue
        depth, i, start = 1, m.end(), m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        inner = text[start:i - 1]
        n = len(_split_args(inner)) if inner.strip() else 0
        assert n == protos[name], f"C.{name} called with {n} arguments, the header declares {protos[name]}"
        calls += 1
    assert calls >= 15
    for name in set(re.findall(r"\bC\.(EPPK_\w+)", text)):
        assert name in consts, f"C.{name} is not defined in include/eppk.h"
    # PickResult.Fallbacks is filled (server.go:74), several devices go through the group API, and the index learns on the device
    assert "Fallbacks: r.fallbacks" in text and "C.eppk_pick_topk(" in text and "C.eppk_group_pick_batch(" in text and "C.EPPK_PICK_LEARN" in text
