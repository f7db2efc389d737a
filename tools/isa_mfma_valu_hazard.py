"""Static audit of shipped gfx950 code objects for the MFMA-result hazard behind inline asm (round 6, session 2).

The result of a double-precision MFMA reaches its VGPRs several cycles after issue; an instruction that READS (or rewrites)
one of those registers too early sees the old contents.  The compiler pads the MFMAs it selects itself (LLVM
GCNHazardRecognizer::checkMAIVALUHazards: a DMFMA 4x4 write needs 6 wait states before a VALU read or write of the register,
9 before a memory / LDS / export read; the 16x16 forms 11 / 18, on gfx950 19 / 18) -- it does not know that an inline-asm
statement IS an MFMA, and it may also move ordinary instructions across it.  kernels_nmf5.hip spells the first product's
accumulate chain in asm (VGPR results, `QV`); with more than one partial chain per column group the VALU adds the chains up
right behind the last links, and one instantiation (<10 x 4 components, 2 groups>) read a chain register two cycles after
the MFMA that writes it: every W update of that shape 1e-2 wrong.  This tool lists every

    v_mfma_f64_* with a VGPR destination

whose destination is read or rewritten by a non-MFMA instruction inside the window (an instruction counts one wait state,
s_nop N counts N + 1 -- the hazard recognizer's own arithmetic; a label or a branch ends the window as "nothing seen").
An MFMA that takes the result as an operand (the next link of a chain) is listed when it follows inside four wait states.

    python tools/isa_mfma_valu_hazard.py flucoma-core_amd/lib/libflucoma_hip.so [more libraries or .co / .s files]

Exit status 1 when a hit exists.  tests/test_isa_audit.py runs it over the production library."""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("isa_store_hazard", os.path.join(_HERE, "isa_store_hazard.py"))
_sh = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_sh)
code_objects, vregs, OBJDUMP = _sh.code_objects, _sh.vregs, _sh.OBJDUMP

DMFMA = re.compile(r"^v_mfma_f64_(4x4x4|16x16x4)")
# (VALU read / write, memory-class read) wait states behind the write of the result
NEED = {"4x4x4": (6, 9), "16x16x4": (19, 18)}
NEED_MFMA = 4
MEMCLASS = ("ds_", "global_", "flat_", "buffer_", "scratch_", "exp")


def operands(ins):
    name, _, rest = ins.partition(" ")
    ops = []
    for o in rest.split(","):
        o = o.strip().split(" ")[0]          # (modifiers like `offset:16`, `sc1` follow the last operand after a space)
        if o:
            ops.append(o)
    return name, ops


def reads_and_writes(ins):
    """(registers read, registers written) of a non-MFMA instruction, as far as VGPRs go"""
    name, ops = operands(ins)
    regs = [vregs(o.lstrip("-|").rstrip("|")) for o in ops]
    if not regs:
        return set(), set()
    stores = name.startswith(("ds_write", "ds_add", "ds_sub", "ds_max", "ds_min", "ds_or", "ds_and", "ds_xor")) or "_store" in name or name.startswith("exp")
    if stores or name.startswith(("v_cmp", "s_")):
        return set().union(*regs), set()
    rd = set().union(*regs[1:]) if len(regs) > 1 else set()
    wr = regs[0]
    if name.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_accvgpr_write")) or "_dpp" in ins or "dpp" in name:
        rd = rd | wr                         # (the destination is also an input)
    return rd, wr


def audit_text(text, label):
    hits = []
    kernel = "?"
    lines = []
    for raw in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw.strip())
        if m:
            kernel = m.group(1)
            lines.append(("label", kernel))
            continue
        s = raw.split("//")[0].split(";")[0].strip()
        if not s or s.endswith(":") and not s.startswith(("v_", "s_", "ds_", "global_", "flat_", "buffer_")):
            if s.endswith(":"):
                if s.startswith("_Z"):
                    kernel = s[:-1]
                lines.append(("label", s))
            continue
        if s.startswith("."):
            continue
        lines.append(("ins", s, kernel))
    for i, item in enumerate(lines):
        if item[0] != "ins":
            continue
        ins = item[1]
        m = DMFMA.match(ins)
        if not m:
            continue
        _, ops = operands(ins)
        dst = vregs(ops[0]) if ops else set()
        if not dst:
            continue                         # (AGPR results: the compiler's own instructions, read through v_accvgpr_read)
        need_valu, need_mem = NEED[m.group(1)]
        ws = 0
        j = i + 1
        while j < len(lines) and ws < max(need_valu, need_mem):
            nxt = lines[j]
            if nxt[0] == "label":
                break
            t = nxt[1]
            if t.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                break
            if t.startswith(("v_mfma", "v_smfmac")):
                # the next link of an accumulate chain (or any MFMA that takes the result as an operand): LLVM asks four wait
                # states between a DMFMA 4x4 write and an overlapping MFMA read; the asm chains keep their links NG x PP >= 5
                # instructions apart by construction -- held here
                _, mops = operands(t)
                srcs = set().union(*[vregs(o) for o in mops[1:]]) if len(mops) > 1 else set()
                if (srcs & dst) and ws < NEED_MFMA:
                    hits.append((label, item[2], ins, t, ws, NEED_MFMA))
                    break
            else:
                rd, wr = reads_and_writes(t)
                mem = t.startswith(MEMCLASS)
                touched = rd if mem else (rd | wr)     # (a load's destination arrives long after the MFMA has retired)
                if (touched & dst) and ws < (need_mem if mem else need_valu):
                    hits.append((label, item[2], ins, t, ws, need_mem if mem else need_valu))
                    break
            mm = re.match(r"s_nop (\d+)", t)
            ws += (int(mm.group(1)) + 1) if mm else 1
            j += 1
    return hits


def audit_file(path):
    hits = []
    if path.endswith(".s"):
        return audit_text(open(path).read(), path), 1
    objs = code_objects(path)
    for k, blob in enumerate(objs):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            r = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--no-show-raw-insn", f.name], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-500:])
            hits += audit_text(r.stdout, f"{os.path.basename(path)}#{k}")
    return hits, len(objs)


def main():
    bad = 0
    for p in sys.argv[1:]:
        hits, n = audit_file(p)
        print(f"{p}: {n} code object(s), {len(hits)} double-precision MFMA results touched inside their hazard window")
        for h in hits[:16]:
            print(f"  {h[1][:100]}\n      {h[2]}\n      {h[3]}   ({h[4]} wait states behind the MFMA, {h[5]} needed)")
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
