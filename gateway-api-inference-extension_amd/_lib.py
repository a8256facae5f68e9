"""ctypes binding of libeppk.so (include/eppk.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os
import sys

EPPK_MAX_SCORERS = 8
EPPK_MAX_PODS = 4096
EPPK_MAX_ADAPTERS = 128
EPPK_MAX_BLOCKS = 256
EPPK_NO_PICK = -1

STATUS = {0: "EPPK_OK", -1: "EPPK_ERR_ARG", -2: "EPPK_ERR_LIMIT", -3: "EPPK_ERR_DEVICE",
          -4: "EPPK_ERR_NO_SNAPSHOT", -5: "EPPK_ERR_INDEX_FULL", -6: "EPPK_ERR_NOMEM"}

# every symbol include/eppk.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "eppk_abi_version", "eppk_create", "eppk_destroy", "eppk_last_error",
    "eppk_snapshot_publish", "eppk_snapshot_info",
    "eppk_index_clear", "eppk_index_insert", "eppk_index_insert_picks_device", "eppk_index_remove_pod",
    "eppk_index_size", "eppk_index_dropped", "eppk_index_selfcheck", "eppk_stream_wait_pick", "eppk_index_advance_epoch", "eppk_index_evict_older",
    "eppk_index_evict_older_device", "eppk_index_trim_pods",
    "eppk_pick_batch", "eppk_pick_batch_device", "eppk_pick_learn_device", "eppk_pick_topk", "eppk_pick_topk_device",
    "eppk_hash_prompt", "eppk_hash_prompts_device", "eppk_xxh64", "eppk_subset_mask", "eppk_round_robin",
    "eppk_addr_fingerprint", "eppk_subset_entries", "eppk_snapshot_set_addresses", "eppk_subset_masks_device", "eppk_subset_masks",
    "eppk_pick_batch_subset", "eppk_pick_batch_candidates_device",
    "eppk_launch_status", "eppk_pick_random_topk", "eppk_pick_random_topk_device", "eppk_set_assumed_load",
    "eppk_group_create", "eppk_group_destroy", "eppk_group_last_error", "eppk_group_size", "eppk_group_ctx", "eppk_group_ranks_seen",
    "eppk_group_set_min_shard", "eppk_group_snapshot_publish", "eppk_group_index_clear", "eppk_group_index_insert",
    "eppk_group_index_remove_pod", "eppk_group_index_advance_epoch", "eppk_group_index_evict_older", "eppk_group_pick_batch",
    "eppk_group_device_picks", "eppk_group_pick_device", "eppk_group_sync", "eppk_group_stream",
    "eppk_group_index_evict_older_device", "eppk_group_index_trim_pods", "eppk_group_pick_topk", "eppk_group_pick_random_topk",
    "eppk_group_pick_stage_buffers", "eppk_group_pick_stage_begin", "eppk_group_pick_stage_end",
    "eppk_host_staging", "eppk_pick_batch_staged", "eppk_pick_stage_buffers", "eppk_pick_stage_begin", "eppk_pick_stage_end", "eppk_chain_is_fused", "eppk_quad_stats", "eppk_resident_stats", "eppk_profile_enable", "eppk_profile_drain", "eppk_profile_bytes",
]


class EppkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class WeightedScorer(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("weight", C.c_int32)]


class Cfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_pods", C.c_uint32),
                ("max_blocks", C.c_uint32), ("max_batch", C.c_uint32), ("index_slots", C.c_uint32),
                ("n_scorers", C.c_uint32), ("reserved", C.c_uint32),
                ("chain", WeightedScorer * EPPK_MAX_SCORERS)]


def lib_path() -> str:
    # EPPK_LIB: development override used by scripts/ab.sh to A/B kernel variants
    return os.environ.get("EPPK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libeppk.so")


_LIB = None


def load_library() -> C.CDLL:
    """Load libeppk.so (built in-tree by ``__graft_entry__.build()``); raise if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the pick.")
    # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7) but links it by the
    # unversioned name: if libeppk brought in /opt/rocm's copy first, a later `import torch` would load a second runtime
    # that finds no GPU ("No HIP GPUs are available").  With torch imported first, libeppk binds to the copy torch loaded.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    lib.eppk_abi_version.restype = u32
    lib.eppk_create.argtypes = [C.POINTER(Cfg), C.POINTER(vp)]
    lib.eppk_destroy.argtypes = [vp]
    lib.eppk_destroy.restype = None
    lib.eppk_last_error.argtypes = [vp]
    lib.eppk_last_error.restype = C.c_char_p
    lib.eppk_snapshot_publish.argtypes = [vp, vp, u32, u64]
    lib.eppk_snapshot_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u64)]
    lib.eppk_index_clear.argtypes = [vp]
    lib.eppk_index_insert.argtypes = [vp, vp, vp, u32]
    lib.eppk_index_insert_picks_device.argtypes = [vp, vp, vp, u32, vp]
    lib.eppk_index_remove_pod.argtypes = [vp, u32]
    lib.eppk_index_size.argtypes = [vp, C.POINTER(u32)]
    lib.eppk_index_dropped.argtypes = [vp, C.POINTER(u64)]
    lib.eppk_index_selfcheck.argtypes = [vp, C.POINTER(u64)]
    lib.eppk_stream_wait_pick.argtypes = [vp, vp]
    lib.eppk_index_advance_epoch.argtypes = [vp, C.POINTER(u32)]
    lib.eppk_index_evict_older.argtypes = [vp, u32, C.POINTER(u32)]
    lib.eppk_index_evict_older_device.argtypes = [vp, u32, vp]
    lib.eppk_index_trim_pods.argtypes = [vp, u32, C.POINTER(u64)]
    lib.eppk_pick_batch.argtypes = [vp, vp, u32, vp, vp, vp]
    lib.eppk_pick_batch_device.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    lib.eppk_pick_learn_device.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    lib.eppk_pick_topk.argtypes = [vp, vp, u32, vp, u32, vp, vp]
    lib.eppk_pick_topk_device.argtypes = [vp, vp, u32, vp, u32, vp, vp, vp]
    lib.eppk_hash_prompt.argtypes = [vp, C.c_size_t, vp, C.c_size_t, u32, vp, u32]
    lib.eppk_hash_prompts_device.argtypes = [vp, vp, u64, vp, vp, vp, u32, u32, vp, vp]
    lib.eppk_xxh64.argtypes = [vp, C.c_size_t, u64]
    lib.eppk_xxh64.restype = u64
    lib.eppk_subset_mask.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u32, C.c_char_p, vp]
    lib.eppk_addr_fingerprint.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, vp]
    lib.eppk_addr_fingerprint.restype = None
    lib.eppk_subset_entries.argtypes = [C.c_char_p, vp, u32]
    lib.eppk_snapshot_set_addresses.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u32]
    lib.eppk_subset_masks_device.argtypes = [vp, vp, vp, u32, vp, vp]
    lib.eppk_subset_masks.argtypes = [vp, vp, vp, u32, vp]
    lib.eppk_pick_batch_subset.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    lib.eppk_pick_batch_candidates_device.argtypes = [vp, vp, u32, vp, u32, vp, vp, vp]
    lib.eppk_round_robin.argtypes = [C.POINTER(u64), u32]
    lib.eppk_round_robin.restype = i32
    lib.eppk_launch_status.argtypes = [vp, C.POINTER(u32)]
    lib.eppk_pick_random_topk.argtypes = [vp, vp, u32, vp, u32, u64, vp, vp]
    lib.eppk_pick_random_topk_device.argtypes = [vp, vp, u32, vp, u32, u64, vp, vp, vp]
    lib.eppk_set_assumed_load.argtypes = [vp, u32]
    lib.eppk_group_create.argtypes = [C.POINTER(Cfg), C.POINTER(i32), u32, u32, C.POINTER(vp)]
    lib.eppk_group_destroy.argtypes = [vp]
    lib.eppk_group_destroy.restype = None
    lib.eppk_group_last_error.argtypes = [vp]
    lib.eppk_group_last_error.restype = C.c_char_p
    lib.eppk_group_size.argtypes = [vp]
    lib.eppk_group_size.restype = u32
    lib.eppk_group_ctx.argtypes = [vp, u32]
    lib.eppk_group_ctx.restype = vp
    lib.eppk_group_ranks_seen.argtypes = [vp]
    lib.eppk_group_set_min_shard.argtypes = [vp, u32]
    lib.eppk_group_snapshot_publish.argtypes = [vp, vp, u32, u64]
    lib.eppk_group_index_clear.argtypes = [vp]
    lib.eppk_group_index_insert.argtypes = [vp, vp, vp, u32]
    lib.eppk_group_index_remove_pod.argtypes = [vp, u32]
    lib.eppk_group_index_advance_epoch.argtypes = [vp, C.POINTER(u32)]
    lib.eppk_group_index_evict_older.argtypes = [vp, u32, C.POINTER(u32)]
    lib.eppk_group_pick_batch.argtypes = [vp, vp, u32, vp, vp, vp, u32]
    lib.eppk_group_index_evict_older_device.argtypes = [vp, u32]
    lib.eppk_group_index_trim_pods.argtypes = [vp, u32, C.POINTER(u64)]
    lib.eppk_group_pick_topk.argtypes = [vp, vp, u32, vp, u32, vp, vp]
    lib.eppk_group_pick_random_topk.argtypes = [vp, vp, u32, vp, u32, u64, vp, vp]
    lib.eppk_group_pick_stage_buffers.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(vp)]
    lib.eppk_group_pick_stage_begin.argtypes = [vp, u32, u32, C.c_int, u32]
    lib.eppk_group_pick_stage_end.argtypes = [vp, u32, vp, vp]
    lib.eppk_group_device_picks.argtypes = [vp, u32]
    lib.eppk_group_device_picks.restype = vp
    lib.eppk_group_pick_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u32]
    lib.eppk_group_sync.argtypes = [vp]
    lib.eppk_group_stream.argtypes = [vp, u32]
    lib.eppk_group_stream.restype = vp
    lib.eppk_chain_is_fused.argtypes = [vp]
    lib.eppk_quad_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.eppk_resident_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.eppk_host_staging.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.eppk_pick_batch_staged.argtypes = [vp, u32, C.c_int, vp, vp]
    lib.eppk_pick_stage_buffers.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(vp)]
    lib.eppk_pick_stage_begin.argtypes = [vp, u32, u32, C.c_int, u32]
    lib.eppk_pick_stage_end.argtypes = [vp, u32, vp, vp]
    lib.eppk_profile_enable.argtypes = [vp, C.c_int]
    lib.eppk_profile_drain.argtypes = [vp, vp, u32, C.POINTER(u32)]
    lib.eppk_profile_bytes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
    _LIB = lib
    return lib
