"""Compile order of a Fortran source set from its `use` statements.

    deporder.py <reference src dir> [<drop-in dir>] [--skip name.f90 ...]

prints "path path ..." (module dependencies first).  With a second directory, every file there replaces the reference file of
the same name and the rest (udc_iface.f90, decomp_2d.f90) is added -- the source set of INTEGRATION.md section 1.  --skip leaves files
out (the one-rank builds take their decomp_2d from standins/ instead of the MPI y-slab module)."""
import glob
import os
import re
import sys


def order(dirs, skip=()):
    files = {}
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "*.f90"))):
            if os.path.basename(f) not in skip:
                files[os.path.basename(f)] = f
    prov, uses = {}, {}
    for f in files.values():
        t = open(f, errors="replace").read()
        for m in re.findall(r"^\s*module\s+(\w+)\s*$", t, re.M | re.I):
            prov[m.lower()] = f
        uses[f] = set(m.lower() for m in re.findall(r"^\s*use\s+(\w+)", t, re.M | re.I))
    out, seen = [], set()

    def visit(f, stack=()):
        if f in seen:
            return
        if f in stack:
            raise SystemExit("dependency cycle through " + f)
        for u in sorted(uses[f]):
            g = prov.get(u)
            if g and g != f:
                visit(g, stack + (f,))
        seen.add(f)
        out.append(f)

    for f in sorted(files.values()):
        visit(f)
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    skip = ()
    if "--skip" in args:
        q = args.index("--skip")
        args, skip = args[:q], tuple(args[q + 1:])
    print(" ".join(order(args, skip)))
