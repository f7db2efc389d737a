"""Static audit of shipped gfx950 code objects for the store-data hazard behind inline asm (round 5).

gfx940-class parts read the data registers of a store of more than 64 bits over several cycles after issue; a VALU
instruction that writes one of those registers within TWO wait states of the store corrupts the stored data (LLVM
GCNHazardRecognizer: the "VMEM store data" hazard, 2 wait states with gfx940 instructions).  The compiler inserts the
wait states behind the stores it emits itself but does not look inside inline-asm statements -- kernels_nmf5.hip's
write-through result store is one.  This tool takes the device code out of a shared library (the clang offload bundles in
its .hip_fatbin section), disassembles every gfx950 code object and reports every

    global_ / flat_ / buffer_ / scratch_ store of 3 or 4 dwords

whose data registers are written by a VALU instruction (v_*, v_accvgpr_read included) fewer than two wait states behind it
(an instruction counts one wait state, s_nop N counts N + 1; a branch target or a branch ends the window conservatively
as "no hazard seen" -- the compiler never schedules across them either).

    python tools/isa_store_hazard.py [--ds] flucoma-core_amd/lib/libflucoma_hip.so [more libraries or .co / .s files]

Exit status 1 when a hit exists.  tests/test_isa_audit.py runs it over the production library.  (--ds also lists wide LDS
stores followed that closely by a write of their data: the compiler itself emits hundreds of those -- the LDS path takes its
data at issue, the hazard is the memory pipeline's -- so they are informational.)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
WIDE_VMEM = re.compile(r"^(global|flat|buffer|scratch)_store_(dwordx3|dwordx4|b96|b128)\b")
WIDE_DS = re.compile(r"^ds_write_(b96|b128)\b|^ds_write2(st64)?_b64\b")
NEED = 2


def code_objects(path):
    """gfx950 ELF images inside a host shared library (or the file itself when it is a code object)"""
    data = open(path, "rb").read()
    if data[:4] == b"\x7fELF" and MAGIC not in data:
        return [data]
    out = []
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def vregs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def store_data(ins):
    """data registers of a wide store instruction (operand text as the disassembler prints it)"""
    name, _, rest = ins.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if name.startswith("ds_write2"):
        return vregs(ops[1]) | vregs(ops[2])
    if name.startswith("ds_write"):
        return vregs(ops[1])
    if name.startswith("buffer_"):
        return vregs(ops[0])
    return vregs(ops[1])          # global / flat / scratch: vaddr, vdata, saddr


def valu_dst(ins):
    name, _, rest = ins.partition(" ")
    if not name.startswith("v_") or name.startswith("v_cmp") and not name.startswith("v_cmpx"):
        # (v_cmp writes VCC / SGPRs; v_cmpx writes EXEC)
        return set()
    if name.startswith("v_mfma") or name.startswith("v_smfmac"):
        return set()              # (matrix results: a different hazard class, handled by the compiler's MAI tables)
    ops = [o.strip() for o in rest.split(",")]
    return vregs(ops[0]) if ops else set()


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
                    kernel = s[:-1]          # (compiler assembly: the symbol line of a kernel)
                lines.append(("label", s))
            continue
        if s.startswith("."):
            continue
        lines.append(("ins", s, kernel))
    for i, item in enumerate(lines):
        if item[0] != "ins":
            continue
        ins = item[1]
        vm, ds = WIDE_VMEM.match(ins), WIDE_DS.match(ins)
        if not (vm or ds):
            continue
        data = store_data(ins)
        if not data:
            continue
        ws = 0
        j = i + 1
        while j < len(lines) and ws < NEED:
            nxt = lines[j]
            if nxt[0] == "label":
                break
            t = nxt[1]
            if t.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                break
            if valu_dst(t) & data:
                hits.append(("vmem" if vm else "ds", label, item[2], ins, t, ws))
                break
            m = re.match(r"s_nop (\d+)", t)
            ws += (int(m.group(1)) + 1) if m else 1
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
    args = [a for a in sys.argv[1:] if a != "--ds"]
    for p in args:
        hits, n = audit_file(p)
        vm = [h for h in hits if h[0] == "vmem"]
        ds = [h for h in hits if h[0] == "ds"]
        print(f"{p}: {n} code object(s), {len(vm)} wide VMEM stores with a VALU write of their data inside {NEED} wait states"
              + (f", {len(ds)} wide LDS stores likewise (informational)" if "--ds" in sys.argv else ""))
        for h in (vm + (ds if "--ds" in sys.argv else []))[:12]:
            print(f"  [{h[0]}] {h[2][:90]}\n      {h[3]}\n      {h[4]}   ({h[5]} wait states behind the store)")
        bad += len(vm)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
