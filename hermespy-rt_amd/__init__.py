"""hermespy-rt_amd: MI355X-native compute_paths (the hermespy-rt hot path).

    abi      ctypes mirror of the reference's C ABI (drives the product AND the reference .so)
    lib      loader + ctypes bindings of libhermespy_rt_amd.so (fails loudly if not built)
    device   device-resident tracer on torch-owned HBM buffers and streams
    sharding one-process-per-GPU ray sharding + RCCL gather (torch.distributed)

The drop-in Python module of the reference surface is `hermespy_rt` (pybind11), built into
hermespy-rt_amd/lib/.
"""
import os

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
# HRT_LIB_DIR: an instrumented build of the same sources in another directory (profiles/ scripts only)
LIB_DIR = os.environ.get("HRT_LIB_DIR") or os.path.join(PACKAGE_DIR, "lib")
REPO_DIR = os.path.dirname(PACKAGE_DIR)
SCENES_DIR = os.path.join(REPO_DIR, "scenes")

from . import abi  # noqa: E402,F401
