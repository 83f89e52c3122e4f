"""Readable names for the kernels of libchap_hip.so in rocprofv3 output.  Since round 3 every kernel of the networks' path runs behind
the generic trampoline `chap_grouped<Args, &body<...>, MAXT, MINW>` (csrc/launch.h); profilers print it mangled (or demangled at
great length).  short_name() reduces either form to `body<template arguments>`, e.g. conv_fwd_kernel<bf16,3,1,false,16,1,4,false,true,false,true>."""
import re

_TOK = re.compile(r"Li(\d+)E|Lb([01])E|(t)|(f)")


def short_name(name):
    m = re.search(r"EXadL_Z\d+(\w+?_kernel)(?:I((?:t|f|Li\d+E|Lb[01]E)+)E)?", name)      # mangled trampoline
    if m:
        args = []
        for t in _TOK.finditer(m.group(2) or ""):
            args.append(t.group(1) or ({"0": "false", "1": "true"}.get(t.group(2)) if t.group(2) else ("bf16" if t.group(3) else "f32")))
        return m.group(1) + ("<" + ",".join(args) + ">" if args else "")
    m = re.search(r"&?\(?(\w+_kernel(?:<[^()]*?>)?)\(", name.replace("void ", ""))         # demangled
    if m:
        return m.group(1).replace("unsigned short", "bf16").replace(" ", "")
    return name.replace("void ", "").split("(")[0][:120]


if __name__ == "__main__":
    import sys
    for line in sys.stdin:
        print(short_name(line.strip()))
