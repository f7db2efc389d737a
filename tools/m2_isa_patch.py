"""Hand patches of ONE kernel inside the compiler's device assembly of kernels_nmf5.hip, for tools/m2_isa_probe.cpp:

    python tools/m2_isa_patch.py in.s out.s <kernel symbol substring> <patch> [<patch> ...]

Patches (each leaves every other instruction of the kernel where it was):
  none        the file as it is
  init        the accumulators the compiler zeroes with v_accvgpr_mov_b32 from another (zeroed) AGPR: v_accvgpr_write_b32 aN, 0
  bperm_nop   16 wait states behind every ds_bpermute_b32 pair of the epilogue
  bperm_wait  s_waitcnt lgkmcnt(0) behind every ds_bpermute_b32 pair of the epilogue
  store_nop   two wait states behind every global_store_dwordx4 (all of them come from inline asm here, where the compiler's
              hazard recognizer does not see a store of more than 64 bits whose data registers the next VALU may not write)
  bperm_src   (an earlier hypothesis) a VALU instruction that overwrites a ds_bpermute_b32's DATA register right behind it waits for
              the permute to have returned: s_waitcnt lgkmcnt(0) only in front of such an instruction
"""
import re
import sys


def kernel_range(lines, key):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l.split(":")[0] and l.split(":")[0].endswith("E"))
    end = next(j for j in range(start, len(lines)) if lines[j].strip().startswith(".Lfunc_end"))
    return start, end


def regs(tok):
    """register numbers a VGPR operand token names: v12 -> {12}, v[12:13] -> {12, 13}"""
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"-?\|?v\[(\d+):(\d+)\]\|?", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    src, dst, key, patches = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
    lines = open(src).read().split("\n")
    s, e = kernel_range(lines, key)
    body = lines[s:e]
    n = {p: 0 for p in patches}
    out = []
    i = 0
    while i < len(body):
        l = body[i]
        st = l.strip()
        if "init" in patches:
            m = re.match(r"\s*v_accvgpr_mov_b32 (a\d+), a\d+\s*$", l)
            # only the zero-initialisation block in front of the loop copies from a freshly zeroed register; the epilogue's
            # moves (register shuffles of live values) stay
            if m and any("v_accvgpr_write_b32" in body[j] and ", 0" in body[j] for j in range(max(0, i - 3), i)):
                out.append(f"\tv_accvgpr_write_b32 {m.group(1)}, 0")
                n["init"] += 1
                i += 1
                continue
        out.append(l)
        if "store_nop" in patches and st.startswith("global_store_dwordx4"):
            out.append("\ts_nop 1")
            n["store_nop"] += 1
        if st.startswith("ds_bpermute_b32") and not body[i + 1].strip().startswith("ds_bpermute_b32"):
            if "bperm_nop" in patches:
                out += ["\ts_nop 7", "\ts_nop 7"]
                n["bperm_nop"] += 1
            if "bperm_wait" in patches:
                out.append("\ts_waitcnt lgkmcnt(0)")
                n["bperm_wait"] += 1
            if "bperm_src" in patches:
                # data registers of the permutes just issued (this one and the one in front of it, if it is one)
                data = set()
                for j in (i, i - 1):
                    t = body[j].strip()
                    if t.startswith("ds_bpermute_b32"):
                        ops = [x.strip() for x in t[len("ds_bpermute_b32"):].split(",")]
                        if regs(ops[0]) != regs(ops[2]):     # (in place: the return itself is the write)
                            data |= regs(ops[2])
                # the next few instructions: does a VALU write one of them before an s_waitcnt lgkmcnt covers the permute?
                k = i + 1
                while k < len(body) and k < i + 12:
                    t = body[k].strip()
                    if t.startswith("s_waitcnt") and "lgkmcnt" in t:
                        break
                    if t.startswith("v_") and data:
                        ops = [x.strip() for x in t.split(None, 1)[1].split(",")] if " " in t else []
                        if ops and regs(ops[0]) & data:
                            # mark: the wait goes right in front of instruction k
                            body[k] = "\ts_waitcnt lgkmcnt(0)\n" + body[k]
                            n["bperm_src"] += 1
                            break
                    k += 1
        i += 1
    lines[s:e] = out
    open(dst, "w").write("\n".join(lines))
    print("patched", key, n)


if __name__ == "__main__":
    main()
