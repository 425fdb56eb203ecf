"""The C-ABI library loads and exports every symbol include/mmgclip_hip.h declares (no compute, no GPU)."""
import ctypes
import os

from mmgclip import _hip


def test_library_exists_and_loads():
    assert os.path.isfile(_hip.LIB_PATH), "run __graft_entry__.build() first"
    lib = _hip.load()
    assert lib.mmg_abi_version() == 5
    assert lib.mmg_target_arch() == b"gfx950"


def test_every_declared_symbol_is_exported():
    protos = _hip.parse_header()
    assert len(protos) >= 10
    lib = ctypes.CDLL(_hip.LIB_PATH)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_every_exported_symbol_is_declared():
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T mmg_" in l}
    declared = set(_hip.parse_header())
    assert exported <= declared, f"exported but undocumented: {sorted(exported - declared)}"


def test_bad_arguments_are_rejected_without_a_gpu():
    # argument validation happens before any HIP call, so this runs on a CPU-only box
    lib = _hip.load()
    rc = lib.mmg_clip_rows_fwd(None, None, None, 8, 8, 100, 0, None, None, None, 0, None)
    assert rc != 0
    assert b"multiple of 32" in lib.mmg_last_error()
