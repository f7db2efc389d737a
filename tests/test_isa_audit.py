"""Static audit of the shipped device code (no GPU): the store-data hazard that inline asm hides from the compiler
(tools/isa_store_hazard.py; DESIGN section 3 "The store hazard behind the inline asm").  Round 5 found the rounds-3/4
"MODE 2 race" to be exactly this: a 128-bit inline-asm store whose second data register the next instruction rewrote."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("isa_store_hazard", os.path.join(ROOT, "tools", "isa_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_wide_store_has_its_data_rewritten_inside_the_hazard_window(fluhip_lib_path):
    t = _tool()
    hits, n = t.audit_file(fluhip_lib_path)
    assert n >= 9, n                                            # one code object per device translation unit
    vm = [h for h in hits if h[0] == "vmem"]
    assert not vm, vm[:4]


def test_the_audit_sees_the_pattern_that_bit():
    """the instruction pair of the round-4 binary (kernels_nmf5.hip <8, 8, in-place> epilogue), and its repaired form"""
    t = _tool()
    bad = "k:\n\tglobal_store_dwordx4 v[0:1], v[168:171], off sc1\n\tv_accvgpr_read_b32 v169, a162\n\ts_endpgm\n"
    good = "k:\n\tglobal_store_dwordx4 v[0:1], v[168:171], off sc1\n\ts_nop 1\n\tv_accvgpr_read_b32 v169, a162\n\ts_endpgm\n"
    one = "k:\n\tglobal_store_dwordx4 v[0:1], v[168:171], off sc1\n\ts_nop 0\n\tv_mov_b32_e32 v171, 0\n\ts_endpgm\n"
    other = "k:\n\tglobal_store_dwordx4 v[0:1], v[168:171], off sc1\n\tv_mov_b32_e32 v172, 0\n\tv_mov_b32_e32 v0, 0\n\ts_endpgm\n"
    assert len(t.audit_text(bad, "x")) == 1 and len(t.audit_text(one, "x")) == 1
    assert not t.audit_text(good, "x") and not t.audit_text(other, "x")   # (address registers and neighbours are free)
