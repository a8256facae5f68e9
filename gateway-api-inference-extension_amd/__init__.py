"""MI355X-native batched endpoint picker — host-side Python mirror of the reference picker seam.

The directory name is fixed by the project layout and is not a Python identifier; load it with
``__graft_entry__.load_package()`` (module alias ``gaie_amd``).

Only the hot path lives here: ``csrc/`` holds the HIP kernels and the C ABI
(``include/eppk.h``); the Python files mirror the reference interface above that ABI:

* ``picker.BatchedPicker``  — batched ``EndpointPicker.Pick`` (pkg/lwepp/handlers/server.go:79-82)
* ``picker.DeviceGroup``    — the same picker over several GPUs behind the C ABI (eppk_group_*)
* ``picker.RoundRobinPicker`` — the reference's only picker (server.go:84-101), the fail-open fallback
* ``picker.subset_mask``    — candidate filter of handleRequestHeaders (request.go:104-133)
* ``distributed``           — request sharding + the all-gather of picks
* ``workload``              — synthetic snapshot/request tables of SURVEY.md §8(d)
* ``metrics``               — snapshot producer: model-server /metrics -> pod rows -> ``BatchedPicker.publish`` (§8(f)-2)

There is no CPU implementation of the pick in this package: without ``libeppk.so`` and a HIP device
every pick raises.
"""
from . import _lib, distributed, metrics, picker, workload  # noqa: F401
from ._lib import EppkError, lib_path, load_library  # noqa: F401
from .picker import BatchedPicker, DeviceGroup, Endpoint, PickResult, RoundRobinPicker, ScorerKind, Unavailable, subset_mask  # noqa: F401
