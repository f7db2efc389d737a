"""Mint tests/golden/param_descriptors.json from the reference's own parameter tables (run where /root/reference exists).

The names, display names, types, defaults, bounds and enum strings of a client's parameters are what a host wrapper builds
its attributes from (`Client::getParameterDescriptors()`, clients/common/FluidNRTClientWrapper.hpp:801-804).  This script
reads the `defineParameters(...)` tables out of the reference headers -- as TEXT, with a small parser; nothing of the
reference is compiled or copied -- and writes them as data, in the shape `tests/cpp/client_driver descriptors` prints the
mirrors' tables (include/flucoma_hip/ParamDescriptors.hpp).  tests/test_client.py compares the two.

    python tools/make_param_descriptor_fixture.py [/root/reference] > tests/golden/param_descriptors.json

Offline clients the reference composes with makeNRTParams (BufMFCC, BufMelBands: rt/MFCCClient.hpp:171-173,
rt/MelBandsClient.hpp:151-153) get the wrapper's parameters in front exactly as FluidNRTClientWrapper.hpp:33-39, :747-785
join them: source, startFrame, numFrames, startChan, numChans, the output buffer, "padding", then the client's table.
NMFFilter / NMFMatch have no offline form upstream; their mirrors put the same wrapper parameters in front of the real-time
client's table (one `resynth` buffer and no padding for the audio-rate NMFFilter, `features` + padding for NMFMatch), so the
fixture composes them that way from the reference's wrapper text and client tables."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
INC = os.path.join(REF, "include", "flucoma", "clients")


def balanced(text, start):
    """text[start] == '(' -> index one past its matching ')'"""
    depth, i, instr = 0, start, False
    while i < len(text):
        ch = text[i]
        if instr:
            if ch == "\\":
                i += 1
            elif ch == '"':
                instr = False
        elif ch == '"':
            instr = True
        elif ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced")


def split_top(body):
    out, depth, cur, instr, angle = [], 0, "", False, 0
    i = 0
    while i < len(body):
        ch = body[i]
        if instr:
            cur += ch
            if ch == "\\":
                cur += body[i + 1]
                i += 1
            elif ch == '"':
                instr = False
        elif ch == '"':
            instr = True
            cur += ch
        elif ch in "(<":
            depth += 1
            cur += ch
        elif ch in ")>":
            depth -= 1
            cur += ch
        elif ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def strip_comments(t):
    return re.sub(r"//[^\n]*", "", t)


def unquote(tok):
    # adjacent string literals ("a" "b") concatenate
    return "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', tok))


def num(tok):
    v = float(tok)
    return int(v) if v == int(v) else v


def parse_entry(entry, index_names, param_names_by_index):
    m = re.match(r"\s*(\w+)\s*(<[^(]*>)?\s*\(", entry)
    kind = m.group(1)
    args = split_top(entry[entry.index("(", m.start(1)) + 1:entry.rindex(")")])
    d = {"name": unquote(args[0]), "display": unquote(args[1])}
    rest = args[2:]
    if kind == "InputBufferParam":
        d["kind"] = "InputBuffer"
    elif kind == "BufferParam":
        d["kind"] = "Buffer"
    elif kind == "FFTParam":
        d["kind"] = "FFT"
        d["default"] = [num(r) for r in rest[:3]]
    elif kind == "EnumParam":
        d["kind"] = "Enum"
        d["default"] = num(rest[0])
        d["strings"] = [unquote(r) for r in rest[1:]]
    elif kind in ("LongParam", "LongParamRuntimeMax", "FloatParam"):
        d["kind"] = "Float" if kind == "FloatParam" else "Long"
        d["default"] = num(rest[0])
        rel = []
        for c in rest[1:]:
            mm = re.match(r"(Min|Max)\(\s*([-\d.]+)\s*\)", c)
            if mm:
                d[mm.group(1).lower()] = num(mm.group(2))
                continue
            mm = re.match(r"(\w+)<(\w+)>\(\)", c)
            if not mm:
                raise ValueError("constraint? " + c)
            rel.append("%s<%s>" % (mm.group(1), param_names_by_index[index_names.index(mm.group(2))]))
        if rel:
            d["relational"] = ", ".join(rel)
    else:
        raise ValueError("parameter kind? " + kind)
    return d


def table(header, which=0):
    """the which-th defineParameters(...) table of a header, with relational constraints resolved through the header's index enum"""
    text = strip_comments(open(os.path.join(INC, header)).read())
    starts = [m.end() - 1 for m in re.finditer(r"defineParameters\s*\(", text)]
    s = starts[which]
    entries = split_top(text[s + 1:balanced(text, s) - 1])
    # the index enum right in front of the table: enumerator i names parameter i
    enums = [m for m in re.finditer(r"enum\s+\w*\s*\{([^}]*)\}", text[:s])]
    index_names = [e.strip().split("=")[0].strip() for e in enums[-1].group(1).split(",") if e.strip()] if enums else []
    names = [unquote(split_top(e[e.index("(") + 1:e.rindex(")")])[0]) for e in entries]
    return [parse_entry(e, index_names, names) for e in entries]


def wrapper_inputs():
    """makeWrapperInputs(B b), clients/common/FluidNRTClientWrapper.hpp:33-39: the four parameters behind the source buffer"""
    text = strip_comments(open(os.path.join(INC, "common", "FluidNRTClientWrapper.hpp")).read())
    s = text.index("makeWrapperInputs(B b)")
    s = text.index("defineParameters", s)
    s = text.index("(", s)
    entries = split_top(text[s + 1:balanced(text, s) - 1])[1:]          # (the first is the forwarded buffer spec)
    return [parse_entry(e, [], []) for e in entries]


def padding_param():
    text = strip_comments(open(os.path.join(INC, "common", "FluidNRTClientWrapper.hpp")).read())
    m = re.search(r'EnumParam\(\s*"padding"', text)
    return parse_entry(text[m.start():balanced(text, text.index("(", m.start()))], [], [])


def nrt_buffers(header):
    """the (input, output) buffer specs a header hands to makeNRTParams"""
    text = strip_comments(open(os.path.join(INC, header)).read())
    m = re.search(r"makeNRTParams<[^>]*>\s*\(", text)
    s = m.end() - 1
    return [parse_entry(e, [], []) for e in split_top(text[s + 1:balanced(text, s) - 1])]


def main():
    win = wrapper_inputs()
    pad = padding_param()
    out = {}
    out["BufNMF"] = table("nrt/NMFClient.hpp")
    out["BufNMFSeed"] = table("nrt/NMFSeedClient.hpp")
    out["BufSTFT"] = table("nrt/BufSTFTClient.hpp")
    for name, header in (("BufMFCC", "rt/MFCCClient.hpp"), ("BufMelBands", "rt/MelBandsClient.hpp")):
        bufs = nrt_buffers(header)
        out[name] = [bufs[0]] + win + bufs[1:] + [pad] + table(header)
    src = {"name": "source", "display": "Source Buffer", "kind": "InputBuffer"}
    out["BufNMFFilter"] = [src] + win + [{"name": "resynth", "display": "Resynthesis Buffer", "kind": "Buffer"}] + table("rt/NMFFilterClient.hpp")
    out["BufNMFMatch"] = [src] + win + [{"name": "features", "display": "Output Buffer", "kind": "Buffer"}, pad] + table("rt/NMFMatchClient.hpp")
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
