"""Snapshot producer, Python mirror: model-server metrics -> the pod rows `BatchedPicker.publish` takes.

The reference's `datastore.Endpoint` carries identity only (pkg/lwepp/datastore/datastore.go:40-46); the gauges the scorers read
are the model-server protocol's (docs/proposals/003-model-server-protocol/README.md:28-57), scraped per endpoint as the data layer
proposal describes (docs/proposals/1023-data-layer-architecture/README.md:106-163: a data source fetches, extractors turn the body
into attributes).  Same behaviour as the C++ twin (host/eppk_metrics.hpp, eppk_scrape.hpp, eppk_producer.hpp) and the Go one in the
patch (gpusnapshot.go); tests/test_metrics_py.py holds the three parsers -- this one, the C++ one, prometheus_client's -- together.
"""
from __future__ import annotations

import math
import threading
import time
import urllib.request
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .picker import POD_DTYPE

POD_INACTIVE = 1          # include/eppk.h: EPPK_POD_INACTIVE
MAX_ADAPTERS = 128        # include/eppk.h: EPPK_MAX_ADAPTERS


@dataclass
class MetricNames:
    """`name` or `name{label=value}` (the Triton forms of 003-…/README.md:30-34); defaults are vLLM's."""
    queued: str = "vllm:num_requests_waiting"
    running: str = "vllm:num_requests_running"
    kv_util: str = "vllm:kv_cache_usage_perc"
    lora_info: str = "vllm:lora_requests_info"


def _spec(s: str) -> Tuple[str, Optional[str], Optional[str]]:
    if "{" not in s:
        return s, None, None
    name, rest = s.split("{", 1)
    k, v = rest.rstrip("}").split("=", 1)
    return name, k, v.strip('"')


def parse_sample(line: str):
    """One line of the Prometheus text format -> (name, labels, value), None for comments / blank lines; ValueError when malformed."""
    line = line.strip(" \t\r")
    if not line or line[0] == "#":
        return None
    i = 0
    while i < len(line) and line[i] not in "{ \t":
        i += 1
    if i == 0:
        raise ValueError("no metric name")
    name, labels = line[:i], {}
    if i < len(line) and line[i] == "{":
        i += 1
        while True:
            while i < len(line) and line[i] in " ,":
                i += 1
            if i < len(line) and line[i] == "}":
                i += 1
                break
            eq = line.find("=", i)
            if eq < 0 or eq + 1 >= len(line) or line[eq + 1] != '"':
                raise ValueError("label without a quoted value")
            key = line[i:eq].rstrip(" ")
            i = eq + 2
            val = []
            while True:
                if i >= len(line):
                    raise ValueError("unterminated label value")
                ch = line[i]
                i += 1
                if ch == "\\" and i < len(line):
                    val.append("\n" if line[i] == "n" else line[i])
                    i += 1
                elif ch == '"':
                    break
                else:
                    val.append(ch)
            labels[key] = "".join(val)
    fields = line[i:].split()
    if not fields:
        raise ValueError("no value")
    v = fields[0]
    value = math.nan if v == "NaN" else math.inf if v in ("+Inf", "Inf") else -math.inf if v == "-Inf" else float(v)
    return name, labels, value


def _count(v: float) -> int:
    return 0 if not v > 0 else 0xFFFFFFFF if v >= 4294967295.0 else int(v + 0.5)


def parse_model_server_metrics(body: str, adapter_ids: Dict[str, int], names: MetricNames = MetricNames()):
    """The body of one model server's /metrics -> (row: np.void of POD_DTYPE, complete: bool, unknown adapter names).
    Several series of one gauge add up (queue, running) or take the maximum (KV utilisation); of the LoRA info series (value =
    timestamp) the latest counts.  complete is False when the queue or the KV gauge is missing."""
    q_spec, r_spec, k_spec = _spec(names.queued), _spec(names.running), _spec(names.kv_util)
    queued = running = kv = 0.0
    has_q = has_kv = False
    lora, stamp = None, -math.inf

    def match(spec, name, labels):
        return name == spec[0] and (spec[1] is None or labels.get(spec[1]) == spec[2])

    for line in body.split("\n"):
        try:
            s = parse_sample(line)
        except ValueError:
            continue
        if s is None or math.isnan(s[2]):
            continue
        name, labels, v = s
        if match(q_spec, name, labels):
            queued, has_q = queued + v, True
        elif match(r_spec, name, labels):
            running += v
        elif match(k_spec, name, labels):
            kv, has_kv = (max(kv, v) if has_kv else v), True
        elif name == names.lora_info and (lora is None or v > stamp):
            lora, stamp = labels, v
    row = np.zeros((), dtype=POD_DTYPE)
    row["queue"], row["running"], row["kv_util"] = _count(queued), _count(running), kv
    unknown: List[str] = []
    if lora is not None:
        try:
            row["max_lora"] = int(lora.get("max_lora", "0"))
        except ValueError:
            pass
        for label, fieldname in (("running_lora_adapters", "active"), ("waiting_lora_adapters", "waiting")):
            bits = [0, 0]
            for a in lora.get(label, "").split(","):
                a = a.strip(" \t")
                if not a:
                    continue
                i = adapter_ids.get(a)
                if i is None or not 0 <= i < MAX_ADAPTERS:
                    if a not in unknown:
                        unknown.append(a)
                    continue
                bits[i >> 6] |= 1 << (i & 63)
            row[fieldname] = bits
    return row, has_q and has_kv, unknown


@dataclass
class ScrapedEndpoint:
    """What the producer lists: identity ("ip:port" is the key, as in the reference) and where the metrics live."""
    address: str
    port: str
    path: str = "/metrics"

    @property
    def key(self) -> str:
        return f"[{self.address}]:{self.port}" if ":" in self.address else f"{self.address}:{self.port}"


@dataclass
class SnapshotProducer:
    """list() -> scrape every endpoint concurrently -> rows -> publish(rows, epoch).  An endpoint keeps its candidate index (slot)
    while it is listed; a slot whose endpoint left or has no usable scrape younger than max_age is published as a hole
    (POD_INACTIVE); trailing holes are dropped.  `publish` is BatchedPicker.publish; `slots` says who sits where."""
    publish: Callable[[np.ndarray, int], None]
    list_endpoints: Callable[[], Sequence[ScrapedEndpoint]]
    names: MetricNames = field(default_factory=MetricNames)
    adapters: Dict[str, int] = field(default_factory=dict)
    timeout_s: float = 1.0
    max_age_s: float = 2.0
    max_pods: int = 4096
    workers: int = 32
    slots: Dict[str, int] = field(default_factory=dict)
    epoch: int = 0
    _latest: Dict[str, Tuple[np.ndarray, float]] = field(default_factory=dict)
    _free: List[int] = field(default_factory=list)
    _n_slots: int = 0

    def _scrape(self, ep: ScrapedEndpoint):
        try:
            host = f"[{ep.address}]" if ":" in ep.address else ep.address
            with urllib.request.urlopen(f"http://{host}:{ep.port}{ep.path}", timeout=self.timeout_s) as r:
                if r.status != 200:
                    return ep, None
                return ep, r.read().decode("utf-8", "replace")
        except Exception:
            return ep, None

    def refresh(self) -> int:
        """One round; returns the number of endpoints in the published snapshot."""
        eps = list(self.list_endpoints())
        with ThreadPoolExecutor(max_workers=max(1, min(self.workers, len(eps) or 1))) as pool:
            results = list(pool.map(self._scrape, eps))
        now = time.monotonic()
        for ep, body in results:
            if body is None:
                continue
            row, complete, unknown = parse_model_server_metrics(body, self.adapters, self.names)
            grew = False
            for a in unknown:                                   # a new adapter name: the lowest free id, then read the sets again
                if a not in self.adapters and len(self.adapters) < MAX_ADAPTERS:
                    self.adapters[a] = min(set(range(MAX_ADAPTERS)) - set(self.adapters.values()))
                    grew = True
            if grew:
                row, complete, _ = parse_model_server_metrics(body, self.adapters, self.names)
            if complete:
                self._latest[ep.key] = (row, now)
        listed = {ep.key for ep in eps}
        for k in [k for k in self._latest if k not in listed]:
            del self._latest[k]
        usable = [ep.key for ep in eps if ep.key in self._latest and now - self._latest[ep.key][1] <= self.max_age_s]
        for k in [k for k in self.slots if k not in usable]:     # leavers free their slots
            self._free.append(self.slots.pop(k))
        for k in usable:                                         # newcomers: lowest free slot, then a fresh one
            if k in self.slots:
                continue
            if self._free:
                self._free.sort()
                self.slots[k] = self._free.pop(0)
            elif self._n_slots < self.max_pods:
                self.slots[k] = self._n_slots
                self._n_slots += 1
        while self._n_slots and (self._n_slots - 1) in self._free:   # trailing holes: shrink
            self._free.remove(self._n_slots - 1)
            self._n_slots -= 1
        rows = np.zeros(self._n_slots, dtype=POD_DTYPE)
        rows["flags"] = POD_INACTIVE
        for k, s in self.slots.items():
            rows[s] = self._latest[k][0]
        self.epoch += 1
        self.publish(rows, self.epoch)
        return len(self.slots)

    def run(self, stop: threading.Event, interval_s: float = 0.05) -> None:
        while not stop.is_set():
            t0 = time.monotonic()
            self.refresh()
            stop.wait(max(0.0, interval_s - (time.monotonic() - t0)))
