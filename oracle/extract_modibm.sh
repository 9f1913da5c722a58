#!/bin/bash
# TEST INFRASTRUCTURE (oracle/_ref build only).
#
# The reference's src/modibm.f90 cannot be compiled here as a whole: initibm and ibmwallfun write the facet statistics
# (lwritefac) through modstat_nc, which needs NetCDF (absent in this image).  Nothing else in the file does -- initfac, which
# the wall functions use, names NetCDF only for the view factors of the energy balance (oracle/extract_initfac.sh).  This
# script assembles, AT BUILD TIME and only into oracle/_ref/, a
# compile unit `module modibm` from the reference file where it lies:
#     :24-128   the module's own declarations (lbottom, point counts, masks, solid_info_type / bound_info_type and their
#               instances) -- minus the `public ::` statement (:30-34), whose list names the routines left out
#     :131-136, :138-197, :249   initibm without its `use modstat_nc` (:137) and without the facet-statistics file it opens
#               under lwritefac (:199-248)
#     :252-270  initibmnorm (reads a solid_*.txt list with the reference's read_sparse_ijk)
#     :273-694  initibmwallfun (facet sections, reconstruction points), plane_line_intersection
#     :697-745  ibmnorm
#     :748-826  solid
#     :889-1164 advecc2nd_corr_conservative, advecc2nd_corr_liberal, diffu_corr, diffv_corr, diffw_corr, diffc_corr
#     :1167-1172, :1174-1245, :1283   ibmwallfun without its `use modstat_nc` (:1173) and the facet statistics (:1246-1282)
#     :1286-1433 wallfunmom, :1436-1607 wallfunheat
#     :1610-1995 trilinear interpolation, alignment, local_coords, the transfer coefficients, moist_flux
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
  sed -n '131,136p' "$SRC"
  sed -n '138,197p' "$SRC"
  sed -n '249p' "$SRC"
  sed -n '252,270p' "$SRC"
  sed -n '273,694p' "$SRC"
  sed -n '697,745p' "$SRC"
  sed -n '748,826p' "$SRC"
  sed -n '889,1164p' "$SRC"
  sed -n '1167,1172p' "$SRC"
  sed -n '1174,1245p' "$SRC"
  sed -n '1283p' "$SRC"
  sed -n '1286,1433p' "$SRC"
  sed -n '1436,1607p' "$SRC"
  sed -n '1610,1995p' "$SRC"
  sed -n '1998,2100p' "$SRC"
  sed -n '2103,2236p' "$SRC"
} > "$OUT"
grep -q "subroutine createmasks" "$OUT" && grep -q "subroutine diffc_corr" "$OUT" && grep -q "subroutine bottom" "$OUT"
