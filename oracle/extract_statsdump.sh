#!/bin/bash
# TEST INFRASTRUCTURE (oracle/_ref build only).
#
# The reference's src/modstatsdump.f90 cannot be compiled here as a whole: initstatsdump and the output half of statsdump
# go through modstat_nc, which needs NetCDF (absent in this image); the tree and TKE-budget dumps pull in `vegetation` and
# `modstatistics`.  The sampling half of statsdump -- interpolations to the flux points, the SGS fluxes, the slab averages,
# the running time averages held in modfields -- and the final slab averages of xytdump do not.  This script assembles, AT
# BUILD TIME and only into oracle/_ref/, a compile unit `module modstatsdump` from the reference file where it lies:
#     :27-32, :35-68   module statement, uses, the module's own variables (clocks tsamplep / tstatsdumpp), `contains`
#                      -- minus `private` / `PUBLIC ::` (:33-34), whose list names exitstatsdump, which is left out
#     :74, :504-508    initstatsdump reduced to its last statements: both clocks to zero
#     :514-534, :536-538, :541-1235   statsdump: uses (minus modstat_nc :535, modstatistics :539, vegetation :540),
#                      declarations, the sampling block up to and including the running averages of the 3-D fields
#     :1291            closes `if (lytdump .or. ... )` (what lies between is the tree dump and commented-out text)
#     :1394-1402       the sample clock, `if (tstatsdumpp >= tstatsdump) then`
#     :1404-1431, :1463   xytdump's final slab averages (the NetCDF calls and the output table :1434-1462 are left out)
#     :1723-1736       the dump clock, deallocations, end subroutine
#     :2172            end module
# Nothing is edited inside those ranges and no reference text is stored in the repository.
set -e
SRC=${1:?path to the reference src/modstatsdump.f90}
OUT=${2:?output file}
n=$(wc -l < "$SRC")
[ "$n" -eq 2172 ] || { echo "extract_statsdump.sh: $SRC has $n lines, expected 2172 (line ranges are pinned to this snapshot)" >&2; exit 1; }
{
  sed -n '27,32p' "$SRC"
  sed -n '35,68p' "$SRC"
  sed -n '74p' "$SRC"
  sed -n '504,508p' "$SRC"
  sed -n '514,534p' "$SRC"
  sed -n '536,538p' "$SRC"
  sed -n '541,1235p' "$SRC"
  sed -n '1291p' "$SRC"
  sed -n '1394,1402p' "$SRC"
  sed -n '1404,1431p' "$SRC"
  sed -n '1463p' "$SRC"
  sed -n '1723,1736p' "$SRC"
  sed -n '2172p' "$SRC"
} > "$OUT"
grep -q "subroutine statsdump" "$OUT" && grep -q "tketxyc" "$OUT" && ! grep -q "writestat" "$OUT"
