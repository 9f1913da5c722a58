#!/bin/bash
# TEST INFRASTRUCTURE (oracle/_ref build only).
#
# The reference's src/modibm.f90 cannot be compiled here as a whole: initibm, initibmwallfun, ibmwallfun, wallfunmom and
# wallfunheat pull in initfac and modstat_nc, which need NetCDF (absent in this image).  The routines of the sparse
# immersed-boundary corrections themselves do not.  This script assembles, AT BUILD TIME and only into oracle/_ref/, a
# compile unit `module modibm` from the reference file where it lies:
#     :24-128   the module's own declarations (lbottom, point counts, masks, solid_info_type / bound_info_type and their
#               instances) -- minus the `public ::` statement (:30-34), whose list names the routines left out
#     :252-270  initibmnorm (reads a solid_*.txt list with the reference's read_sparse_ijk)
#     :697-745  ibmnorm
#     :748-826  solid
#     :889-1164 advecc2nd_corr_conservative, advecc2nd_corr_liberal, diffu_corr, diffv_corr, diffw_corr, diffc_corr
#     :1998-2100 bottom
#     :2103-2236 createmasks, end module
# Nothing is edited inside those ranges and no reference text is stored in the repository.
set -e
SRC=${1:?path to the reference src/modibm.f90}
OUT=${2:?output file}
n=$(wc -l < "$SRC")
[ "$n" -eq 2236 ] || { echo "extract_modibm.sh: $SRC has $n lines, expected 2236 (line ranges are pinned to this snapshot)" >&2; exit 1; }
{
  sed -n '24,29p' "$SRC"
  sed -n '35,129p' "$SRC"
  sed -n '252,270p' "$SRC"
  sed -n '697,745p' "$SRC"
  sed -n '748,826p' "$SRC"
  sed -n '889,1164p' "$SRC"
  sed -n '1998,2100p' "$SRC"
  sed -n '2103,2236p' "$SRC"
} > "$OUT"
grep -q "subroutine createmasks" "$OUT" && grep -q "subroutine diffc_corr" "$OUT" && grep -q "subroutine bottom" "$OUT"
