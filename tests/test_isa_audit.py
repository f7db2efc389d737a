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


def _mfma_tool():
    spec = importlib.util.spec_from_file_location("isa_mfma_valu_hazard", os.path.join(ROOT, "tools", "isa_mfma_valu_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_double_precision_mfma_result_is_touched_inside_its_hazard_window(fluhip_lib_path):
    """round 6: the first product's accumulate chain is spelled in asm (VGPR results); the compiler neither pads the VALU reads
    behind an asm MFMA nor keeps them behind it -- one instantiation read a chain register two cycles after its MFMA"""
    t = _mfma_tool()
    hits, n = t.audit_file(fluhip_lib_path)
    assert n >= 9, n
    assert not hits, hits[:4]


def test_the_mfma_audit_sees_the_pattern_that_bit():
    """the instruction pair of kernels_nmf5.hip <10 x 4 components, 2 groups> before the wait states went in, and repaired forms"""
    t = _mfma_tool()
    bad = ("k:\n\tv_mfma_f64_4x4x4_4b_f64 v[126:127], v[108:109], v[20:21], v[126:127]\n\ts_nop 0\n"
           "\tv_add_f64 v[106:107], v[150:151], v[126:127]\n\ts_endpgm\n")
    good = ("k:\n\tv_mfma_f64_4x4x4_4b_f64 v[126:127], v[108:109], v[20:21], v[126:127]\n\ts_nop 11\n"
            "\tv_add_f64 v[106:107], v[150:151], v[126:127]\n\ts_endpgm\n")
    link = "\tv_mfma_f64_4x4x4_4b_f64 v[%d:%d], v[108:109], v[20:21], v[%d:%d]\n"
    chain = ("k:\n" + link % (126, 127, 126, 127) + "".join(link % (r, r + 1, r, r + 1) for r in (128, 130, 132, 134)) + link % (126, 127, 126, 127)
             + "\ts_nop 7\n\tv_add_f64 v[2:3], v[126:127], v[4:5]\n\ts_endpgm\n")      # links five instructions apart, the sum behind wait states
    tight = "k:\n" + link % (126, 127, 126, 127) + link % (128, 129, 128, 129) + link % (126, 127, 126, 127) + "\ts_endpgm\n"
    other = ("k:\n\tv_mfma_f64_4x4x4_4b_f64 v[126:127], v[108:109], v[20:21], v[126:127]\n\tv_add_f64 v[106:107], v[150:151], v[128:129]\n"
             "\tds_read_b128 v[124:127], v89\n\ts_endpgm\n")
    store = "k:\n\tv_mfma_f64_4x4x4_4b_f64 v[126:127], v[108:109], v[20:21], 0\n\ts_nop 6\n\tds_write_b64 v1, v[126:127]\n\ts_endpgm\n"
    agpr = "k:\n\tv_mfma_f64_4x4x4_4b_f64 a[0:1], v[108:109], v[20:21], a[0:1]\n\tv_add_f64 v[0:1], v[0:1], v[2:3]\n\ts_endpgm\n"
    assert len(t.audit_text(bad, "x")) == 1 and len(t.audit_text(store, "x")) == 1      # (a memory-class read needs 9)
    assert len(t.audit_text(tight, "x")) == 1                                            # (a chain's next link two instructions on)
    assert not t.audit_text(good, "x") and not t.audit_text(chain, "x") and not t.audit_text(other, "x") and not t.audit_text(agpr, "x")
