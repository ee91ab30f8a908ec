"""The build-time check of the streaming GEMM's drawn tile id (tools/check_draw_isa.py) on hand-made assembly."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location(
    "check_draw_isa", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_draw_isa.py"))
check_draw_isa = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(check_draw_isa)

_SAFE = """
	s_nop 4
	global_atomic_add v224, v38, v164, s[4:5] sc0
	v_mfma_f32_16x16x32_bf16 v[76:79], v[172:175], v[196:199], v[132:135]
	global_load_lds_dwordx4 v238, s[26:27]
	s_waitcnt vmcnt(22)
	ds_read_b128 v[14:17], v225
	s_waitcnt vmcnt(6)
	v_mov_b32_e32 v39, v224
"""


def test_register_left_alone_until_the_wait_passes():
    assert check_draw_isa.check(_SAFE) == []


@pytest.mark.parametrize("bad", [
    "	scratch_store_dword off, v224, off      ; 4-byte Folded Spill",   # spilled while the data is in flight
    "	v_mov_b32_e32 v39, v224",                                          # copied early
    "	v_xor_b32_e32 v224, 0x10000, v215",                                # reused as a temporary
    "	ds_read_b128 v[222:225], v39",                                     # inside a register range
])
def test_early_use_of_the_destination_is_reported(bad):
    asm = _SAFE.replace("	s_waitcnt vmcnt(22)\n", bad + "\n	s_waitcnt vmcnt(22)\n")
    problems = check_draw_isa.check(asm)
    assert len(problems) == 1 and "v224" in problems[0]


def test_a_wait_that_leaves_the_atomic_in_flight_does_not_count():
    asm = _SAFE.replace("	s_waitcnt vmcnt(6)\n", "")
    assert check_draw_isa.check(asm)  # only vmcnt(22) before the use: not covered


def test_assembly_without_the_atomic_is_rejected():
    with pytest.raises(RuntimeError):
        check_draw_isa.check("	s_endpgm\n")
