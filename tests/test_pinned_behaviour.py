"""CPU: the behaviour the reference DOES pin — round-robin picking and the subset filter — restated
case by case from pkg/lwepp/handlers/request_test.go and checked on both the oracle and libeppk's
host functions (through the C ABI)."""
import ctypes as C

import numpy as np
import pytest


def eps(pkg, *pairs):
    return [pkg.picker.Endpoint(a, p) for a, p in pairs]


def cands(pkg, orc, endpoints, filt):
    """Candidate addresses per the library and per the oracle (must agree)."""
    mask, n = pkg.picker.subset_mask(endpoints, filt)
    omask, on = orc.subset_mask([e.address for e in endpoints], [e.port for e in endpoints], filt)
    assert n == on and np.array_equal(mask, omask)
    sel = [e for i, e in enumerate(endpoints) if (int(mask[i >> 6]) >> (i & 63)) & 1]
    assert len(sel) == n
    return sel


def test_round_robin_alternates_and_wraps(pkg, orc):
    # request_test.go:50-88 — two pods: req1 != req2, req3 == req1
    endpoints = eps(pkg, ("10.0.0.1", "8080"), ("10.0.0.2", "8080"))
    rr = pkg.picker.RoundRobinPicker()
    r1, r2, r3 = (rr.Pick(None, endpoints).endpoint for _ in range(3))
    assert r1 != r2 and r3 == r1
    # server.go:95-96 — pre-increment: the first pick on a fresh counter is index 1 % len
    assert r1 == "10.0.0.2:8080"
    ctr = C.c_uint64(0)
    assert [orc.round_robin(ctr, 2) for _ in range(4)] == [1, 0, 1, 0]
    ctr = C.c_uint64(2**64 - 1)                       # uint64 wrap-around of the counter
    assert orc.round_robin(ctr, 3) == 0 and ctr.value == 0


def test_round_robin_no_endpoints_is_unavailable(pkg, orc):
    # server.go:91-93 / request_test.go:117-130
    with pytest.raises(pkg.picker.Unavailable):
        pkg.picker.RoundRobinPicker().Pick(None, [])
    assert orc.round_robin(C.c_uint64(0), 0) == -1


def test_filter_via_header_single_ip(pkg, orc):
    # request_test.go:90-115 — header "10.0.0.2" selects exactly that pod
    e = eps(pkg, ("10.0.0.1", "8080"), ("10.0.0.2", "8080"), ("10.0.0.3", "8080"))
    assert [x.address for x in cands(pkg, orc, e, "10.0.0.2")] == ["10.0.0.2"]


def test_filter_via_metadata_list_and_string(pkg, orc):
    # request_test.go:132-203
    e = eps(pkg, ("10.0.0.1", "8080"), ("10.0.0.2", "8080"), ("10.0.0.3", "8080"))
    assert [x.address for x in cands(pkg, orc, e, "10.0.0.3:8080")] == ["10.0.0.3"]
    assert [x.address for x in cands(pkg, orc, e, "10.0.0.2:8080,10.0.0.3:8080")] == ["10.0.0.2", "10.0.0.3"]


def test_no_subset_returns_all_pods(pkg, orc):
    # request_test.go:281-333 (no metadata / unrelated key) — filter absent
    e = eps(pkg, ("10.0.0.1", "8080"), ("10.0.0.2", "8080"))
    assert len(cands(pkg, orc, e, None)) == 2


def test_empty_or_non_matching_subset_fails_closed(pkg, orc):
    # request_test.go:335-369 (empty list) and :407-439 (no match): zero candidates, never fail open
    e = eps(pkg, ("10.0.0.1", "8080"), ("10.0.0.2", "8080"))
    assert cands(pkg, orc, e, "") == []
    assert cands(pkg, orc, e, "192.168.1.1:8080") == []


def test_whitespace_and_mixed_list_elements(pkg, orc):
    # request_test.go:371-405 and :441-479
    e = eps(pkg, ("10.0.0.1", "8080"), ("10.0.0.2", "8080"), ("10.0.0.3", "8080"))
    assert [x.address for x in cands(pkg, orc, e, "  10.0.0.2:8080 ,  ,  10.0.0.3:8080  ")] == ["10.0.0.2", "10.0.0.3"]
    assert [x.address for x in cands(pkg, orc, e, "10.0.0.2:8080, 10.0.0.3:8080")] == ["10.0.0.2", "10.0.0.3"]


@pytest.mark.parametrize("filt,want", [
    ("10.0.0.1:8080", [("10.0.0.1", "8080")]),                                              # specific port only
    ("10.0.0.1", [("10.0.0.1", "8080"), ("10.0.0.1", "9090")]),                             # ip-only: all ports
    ("10.0.0.1:8080,10.0.0.2", [("10.0.0.1", "8080"), ("10.0.0.2", "8080"), ("10.0.0.2", "9090")]),
    ("10.0.0.1:3000", []),                                                                  # no matching port
])
def test_port_aware_filtering_table(pkg, orc, filt, want):
    # request_test.go:481-551
    e = eps(pkg, ("10.0.0.1", "8080"), ("10.0.0.1", "9090"), ("10.0.0.2", "8080"), ("10.0.0.2", "9090"))
    assert [(x.address, x.port) for x in cands(pkg, orc, e, filt)] == want


def test_split_host_port_edge_cases(pkg, orc):
    # net.SplitHostPort acceptance as used at request.go:110: bracketed IPv6 parses, bare IPv6 is ip-only
    e = eps(pkg, ("::1", "80"), ("::1", "81"), ("fe80::2", "80"))
    assert [(x.address, x.port) for x in cands(pkg, orc, e, "[::1]:80")] == [("::1", "80")]
    assert [(x.address, x.port) for x in cands(pkg, orc, e, "::1")] == [("::1", "80"), ("::1", "81")]
    assert cands(pkg, orc, e, "[::1]") == [] and cands(pkg, orc, e, "[::1]x:80") == []


def test_pick_result_endpoint_format(pkg):
    # server.go:98-100 / server_test.go:59-121 — exact "ip:port" bytes; IPv6 is bracketed by JoinHostPort
    assert pkg.picker.join_host_port("10.0.0.1", "8080") == "10.0.0.1:8080"
    assert pkg.picker.join_host_port("::1", "80") == "[::1]:80"


def test_many_pods_mask_words(pkg, orc):
    e = [pkg.picker.Endpoint(f"10.0.{i // 256}.{i % 256}", "8000") for i in range(200)]
    sel = cands(pkg, orc, e, "10.0.0.5,10.0.0.130:8000, 10.0.0.199:9")
    assert [x.address for x in sel] == ["10.0.0.5", "10.0.0.130"]


def _go_split_host_port(hp):
    """net.SplitHostPort (Go standard library), restated: (host, port) or None on any of its errors."""
    i = hp.rfind(":")
    if i < 0:
        return None                                   # missing port in address
    j = k = 0
    if hp.startswith("["):
        end = hp.find("]")
        if end < 0:
            return None                               # missing ']' in address
        if end + 1 == len(hp):
            return None                               # missing port
        if end + 1 != i:
            return None                               # either "too many colons" or "missing port"
        host = hp[1:end]
        j, k = 1, end + 1
    else:
        host = hp[:i]
        if ":" in host:
            return None                               # too many colons in address
    if "[" in hp[j:] or "]" in hp[k:]:
        return None                                   # unexpected '[' / ']'
    return host, hp[i + 1:]


_GO_SPACE = " \t\n\v\f\r\x85\xa0\u1680" + "".join(chr(c) for c in range(0x2000, 0x200B)) + "\u2028\u2029\u202f\u205f\u3000"   # unicode.IsSpace


def _filter_restated(endpoints, value):
    """request.go:104-133 in Python: entries split on ',', trimmed; host:port entries allow that port, anything else is an
    address whose every port is allowed; returns candidate indices."""
    if value is None:
        return list(range(len(endpoints)))
    allowed, allow_all = {}, set()
    for ep in value.split(","):
        ep = ep.strip(_GO_SPACE)
        hp = _go_split_host_port(ep)
        if hp is not None:
            allowed.setdefault(hp[0], set()).add(hp[1])
        else:
            allow_all.add(ep)
    return [i for i, e in enumerate(endpoints) if e.address in allow_all or e.port in allowed.get(e.address, ())]


def test_subset_filter_against_a_restatement_of_the_go_code(pkg, orc):
    """Random subset strings (addresses, ports, brackets, stray colons, blanks, empty entries) through the library, the oracle and a
    Python restatement of request.go:104-133 + net.SplitHostPort: identical candidate sets."""
    rng = np.random.default_rng(2024)
    addrs = ["10.0.0.1", "10.0.0.2", "::1", "fe80::2", "host-a", "[::1]"]
    ports = ["80", "81", "8080", ""]
    pieces = addrs + ports + [":", "::", "[", "]", " ", "\t", "x", "[::1]", "[fe80::2]", "10.0.0.1:80", "[::1]:81", "host-a:8080",
                              "\u00a0", "\u2003", "\u3000", "\x85", "\u200b"]     # Unicode spaces (U+200B is NOT one)
    for _ in range(400):
        n = int(rng.integers(1, 9))
        endpoints = [pkg.picker.Endpoint(str(rng.choice(addrs)), str(rng.choice(ports))) for _ in range(n)]
        entries = []
        for _ in range(int(rng.integers(0, 5))):
            entries.append("".join(str(rng.choice(pieces)) for _ in range(int(rng.integers(0, 4)))))
        value = ",".join(entries) if rng.random() > 0.05 else None
        want = _filter_restated(endpoints, value)
        cands(pkg, orc, endpoints, value)                      # (asserts library == oracle)
        mask, cnt = pkg.picker.subset_mask(endpoints, value)
        got = [i for i in range(n) if (int(mask[i >> 6]) >> (i & 63)) & 1]
        assert got == want and cnt == len(want), (value, [(e.address, e.port) for e in endpoints])


def test_subset_entries_tokenise_like_the_string_filter(pkg):
    """Host half of the on-device subset filter (include/eppk.h eppk_subset_entries; SEMANTICS.md §5a): the same entry rules as
    eppk_subset_mask -- split on ',', Unicode trim, SplitHostPort decides between an exact and an all-ports entry."""
    import ctypes as C
    lib = pkg._lib.load_library()

    def fp(host, port=None):
        out = np.zeros(2, dtype=np.uint64)
        h = host.encode()
        p = None if port is None else port.encode()
        lib.eppk_addr_fingerprint(h, len(h), p, 0 if p is None else len(p), out.ctypes.data)
        return (int(out[0]), int(out[1]))

    ent = pkg.picker.subset_entries
    assert [tuple(int(x) for x in e) for e in ent(None)] == [(0, 0)]            # no filter: the "every pod" entry
    assert ent("").shape == (0, 2) and ent(" , 　,").shape == (0, 2)         # present but empty: fail closed
    got = [tuple(int(x) for x in e) for e in ent(" 10.0.0.1:80 ,10.0.0.2, [fd00::1]:8080\t,fd00::2,[fd00::3],host:")]
    assert got == [fp("10.0.0.1", "80"), fp("10.0.0.2"), fp("fd00::1", "8080"), fp("fd00::2"), fp("[fd00::3]"), fp("host", "")]
    assert fp("10.0.0.1", "80") != fp("10.0.0.1") and fp("a", "") != fp("a")      # an exact entry never aliases an all-ports one
    # XXH64 under the two seeds of SEMANTICS.md §5a
    assert fp("10.0.0.2") == (lib.eppk_xxh64(b"10.0.0.2", 8, 0), lib.eppk_xxh64(b"10.0.0.2", 8, 0x9E3779B97F4A7C15))
    assert fp("h", "1") == (lib.eppk_xxh64(b"h\x001", 3, 0), lib.eppk_xxh64(b"h\x001", 3, 0x9E3779B97F4A7C15))
    # a filter with more entries than the first buffer
    many = ",".join(f"10.1.{i >> 8}.{i & 255}:{8000 + i}" for i in range(300))
    assert ent(many).shape == (300, 2)
    keys, off = pkg.picker.subset_entries_csr([None, "", many, "10.0.0.2"])
    assert list(off) == [0, 1, 1, 301, 302] and keys.shape == (302, 2)
