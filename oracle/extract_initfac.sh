#!/bin/bash
# TEST INFRASTRUCTURE (oracle/_ref build only).
#
# The reference's src/initfac.f90 (facet properties: normals, roughness lengths, temperatures, read from facets.inp,
# factypes.inp, Tfacinit.inp) names NetCDF in one place: the view factors of the surface energy balance
# (vf.nc.inp, :266-270, inside `if (lEB)`).  Everything the immersed boundary's wall functions use is read from text files.
# This script writes, AT BUILD TIME and only into oracle/_ref/, the reference file without its `use netcdf` (:32) and
# without those five lines; the `else` / `end if` around them stay (an empty branch).  Nothing else is touched and no
# reference text is stored in the repository.  Decks with lEB are refused by the test driver.
set -e
SRC=${1:?path to the reference src/initfac.f90}
OUT=${2:?output file}
n=$(wc -l < "$SRC")
[ "$n" -eq 420 ] || { echo "extract_initfac.sh: $SRC has $n lines, expected 420 (line ranges are pinned to this snapshot)" >&2; exit 1; }
{
  sed -n '1,31p' "$SRC"
  sed -n '33,265p' "$SRC"
  sed -n '271,420p' "$SRC"
} > "$OUT"
grep -q "subroutine readfacetfiles" "$OUT" && ! grep -qi "nf90_\|use netcdf" "$OUT"
