#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/Makefile.ref).  Reads one reference .cu file and writes it to stdout with every kernel
launch `name<targs><<<grid, block[, shmem[, stream]]>>>(` rewritten to `cpu_launch(grid, block, name<targs>, ` so that g++
can compile it against oracle/ref_stub/cuda_runtime.h.  Nothing else is changed and the output is piped straight into the
compiler: no copy of the reference source is kept anywhere."""
import re
import sys


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def rewrite(src):
    out, pos = [], 0
    while True:
        k = src.find("<<<", pos)
        if k < 0:
            out.append(src[pos:]); break
        e = src.index(">>>", k)
        # kernel expression: identifier, optionally followed by a balanced <...> template argument list
        j = k
        if src[j - 1] == ">":
            depth, j = 0, j - 1
            while True:
                if src[j] == ">":
                    depth += 1
                elif src[j] == "<":
                    depth -= 1
                    if depth == 0:
                        break
                j -= 1
        m = re.search(r"[A-Za-z_][A-Za-z_0-9:]*$", src[:j])
        name = src[m.start():k]
        cfg = split_top(src[k + 3:e])
        after = e + 3
        while src[after].isspace():
            after += 1
        assert src[after] == "(", "launch without argument list"
        out.append(src[pos:m.start()])
        out.append("cpu_launch(%s, %s, %s, " % (cfg[0], cfg[1], name))
        pos = after + 1
    return "".join(out)


if __name__ == "__main__":
    sys.stdout.write(rewrite(open(sys.argv[1]).read()))
