"""Runs bench.main() in a subprocess with the CPU test double bound (tests only): checks the output contract of
bench.py — one JSON line with the agreed keys — without a GPU. Numbers from the double mean nothing.
MDHIP_TEST_DIE_RANK=r: the rank with that number dies at once (the self-spawning parent must notice and fail loudly)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("MDHIP_TEST_DIE_RANK") is not None and os.environ.get("RANK") == os.environ["MDHIP_TEST_DIE_RANK"]:
    os._exit(7)
from minidiff_amd import _capi  # noqa: E402

_capi.use_library(os.path.join(ROOT, "oracle", "_build", "libmdhip_host.so"))
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main(entry=os.path.abspath(__file__))
