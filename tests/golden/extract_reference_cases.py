#!/usr/bin/env python3
"""Derive the reference-held vectors for the pinned half of the path from the reference's OWN test file.

Reads  /root/reference/pkg/lwepp/handlers/request_test.go  (Go source; not importable, no Go toolchain here) and
writes tests/golden/reference_cases.json: for every `func TestHandleRequestHeaders_*` the endpoint list the mock
datastore serves, the test header value, the subset-filter metadata (list / string / other key / absent) and every
assertion the Go test makes about candidates and the selected pod.  Nothing is typed by hand: each field is cut out
of the Go text by the patterns below, and the byte range of the source function is recorded so that a reader can
check a case against the file (`source_lines`).

Run (in the build container, where /root/reference exists):
    python tests/golden/extract_reference_cases.py
The GPU box has no /root/reference: tests read the committed JSON only (tests/test_reference_cases.py).
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import sys

REF = "/root/reference/pkg/lwepp/handlers/request_test.go"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_cases.json")

ENDPOINT_RE = re.compile(r'\{\s*Address:\s*"([^"]*)"\s*,\s*Port:\s*"([^"]*)"\s*\}')
STR_RE = re.compile(r'"((?:[^"\\]|\\.)*)"')


def go_unquote(s: str) -> str:
    return bytes(s, "utf-8").decode("unicode_escape")


def match_brace(text: str, open_pos: int) -> int:
    """Index just past the brace that closes text[open_pos] ('{'), skipping string literals and // comments."""
    depth, i, n = 0, open_pos, len(text)
    while i < n:
        c = text[i]
        if c == '"':
            i += 1
            while text[i] != '"':
                i += 2 if text[i] == "\\" else 1
        elif c == "`":
            i = text.index("`", i + 1)
        elif text.startswith("//", i):
            i = text.index("\n", i)
            continue
        elif c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced braces")


def block_after(text: str, marker_re: str, start: int = 0):
    """(body, end) of the first `{...}` that follows a match of marker_re, or (None, start)."""
    m = re.compile(marker_re).search(text, start)
    if not m:
        return None, start
    ob = text.index("{", m.end() - 1)
    end = match_brace(text, ob)
    return text[ob + 1:end - 1], end


def endpoints_in(body: str):
    return [{"address": a, "port": p} for a, p in ENDPOINT_RE.findall(body)]


def parse_metadata(body: str):
    """The value stored under metadata.SubsetFilterKey / any other key of the subset namespace."""
    ns, _ = block_after(body, r"metadata\.SubsetFilterNamespace:\s*\{")
    if ns is None:
        return {"kind": "absent"}
    m = re.search(r"metadata\.SubsetFilterKey:\s*structpb\.(NewListValue|NewStringValue)\(", ns)
    if not m:
        return {"kind": "other-key"}          # the namespace struct exists, the subset key does not
    if m.group(1) == "NewStringValue":
        s = STR_RE.search(ns, m.end())
        val = go_unquote(s.group(1))
        if val == "tc.filterValue":
            raise AssertionError("unreachable")
        return {"kind": "string", "value": val}
    vals, _ = block_after(ns, r"Values:\s*\[\]\*structpb\.Value\{", m.end())
    items = [go_unquote(x) for x in re.findall(r'structpb\.NewStringValue\("((?:[^"\\]|\\.)*)"\)', vals or "")]
    return {"kind": "list", "values": items}


def parse_header(body: str):
    m = re.search(r'\{Key:\s*"test-epp-endpoint-selection",\s*Value:\s*"((?:[^"\\]|\\.)*)"\}', body)
    return go_unquote(m.group(1)) if m else None


def parse_expectations(body: str):
    exp = {}
    m = re.search(r'assert\.Equal\(t,\s*"([^"]*)",\s*reqCtx\.SelectedPodIP\)', body)
    if m:
        exp["selected_ip"] = m.group(1)
    m = re.search(r"assert\.Len\(t,\s*reqCtx\.Candidates,\s*(\d+)\)", body)
    if m:
        exp["n_candidates"] = int(m.group(1))
    m = re.search(r"assert\.ElementsMatch\(t,\s*\[\]string\{([^}]*)\},\s*\[\]string\{reqCtx\.Candidates", body)
    if m:
        exp["candidate_addresses"] = [go_unquote(x) for x in STR_RE.findall(m.group(1))]
    if re.search(r"assert\.Empty\(t,\s*reqCtx\.Candidates\)", body):
        exp["n_candidates"] = 0
    if re.search(r"handleRequestHeaders\([^\n]*\)\s*\n\s*assert\.Error\(t,\s*err\)", body):
        exp["headers_error"] = True           # codes.Unavailable: no pods available (request.go:100-102)
    # round-robin relations between successive picks (request_test.go:50-88)
    rel = []
    for a, b in re.findall(r"assert\.NotEqual\(t,\s*reqCtx(\d)\.SelectedPodIP,\s*reqCtx(\d)\.SelectedPodIP\)", body):
        rel.append({"op": "ne", "a": int(a), "b": int(b)})
    for a, b in re.findall(r"assert\.Equal\(t,\s*reqCtx(\d)\.SelectedPodIP,\s*reqCtx(\d)\.SelectedPodIP\)", body):
        rel.append({"op": "eq", "a": int(a), "b": int(b)})
    if rel:
        exp["pick_relations"] = rel
        exp["n_picks"] = len(set(re.findall(r"reqCtx(\d)\s*:=", body)))
    return exp


def parse_table(body: str):
    """`tests := []struct{...}{ {name:…, filterValue:…, expectedCandidates: …}, … }` (request_test.go:481-551)."""
    m = re.search(r"tests\s*:=\s*\[\]struct\s*\{", body)
    if not m:
        return None
    decl_end = match_brace(body, body.index("{", m.end() - 1))
    rows_open = body.index("{", decl_end)
    rows = body[rows_open + 1:match_brace(body, rows_open) - 1]
    out, i = [], 0
    while True:
        ob = rows.find("{", i)
        if ob < 0:
            break
        end = match_brace(rows, ob)
        row = rows[ob + 1:end - 1]
        name = go_unquote(re.search(r'name:\s*"((?:[^"\\]|\\.)*)"', row).group(1))
        filt = go_unquote(re.search(r'filterValue:\s*"((?:[^"\\]|\\.)*)"', row).group(1))
        expb, _ = block_after(row, r"expectedCandidates:\s*\[\]\*datastore\.Endpoint\{")
        out.append({"name": name, "filter_value": filt, "expected_candidates": endpoints_in(expb or "")})
        i = end
    return out


def main() -> int:
    src = open(REF, encoding="utf-8").read()
    line_of = lambda pos: src.count("\n", 0, pos) + 1   # noqa: E731
    cases = []
    for m in re.finditer(r"^func (Test\w+)\(t \*testing\.T\) \{", src, re.M):
        ob = src.index("{", m.end() - 1)
        end = match_brace(src, ob)
        body = src[ob + 1:end - 1]
        podsb, _ = block_after(body, r"pods\s*:=\s*\[\]\*datastore\.Endpoint\{")
        if podsb is None:
            podsb, _ = block_after(body, r"mockDatastore\{pods:\s*\[\]\*datastore\.Endpoint\{")
        case = {"name": m.group(1), "source_lines": [line_of(m.start()), line_of(end)], "pods": endpoints_in(podsb or "")}
        table = parse_table(body)
        if table is not None:
            case["table"] = table               # the metadata value is tc.filterValue, a string
        else:
            case["header"] = parse_header(body)
            case["metadata"] = parse_metadata(body)
            case["expect"] = parse_expectations(body)
        cases.append(case)
    doc = {
        "generated_by": "tests/golden/extract_reference_cases.py",
        "source": "pkg/lwepp/handlers/request_test.go",
        "source_sha256": hashlib.sha256(src.encode()).hexdigest(),
        "subset_namespace": "envoy.lb.subset_hint",
        "subset_key": "x-gateway-destination-endpoint-subset",
        "cases": cases,
    }
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"{len(cases)} test functions -> {OUT}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
