#!/bin/bash
# TEST INFRASTRUCTURE (oracle/_ref build only).
#
# src/modstartup.f90 cannot be compiled here (it reads the driver / restart input through NetCDF-backed modules), but the
# one routine of it that shapes the cold start -- randomize_field (:2367-2396), the decomposition-independent random
# perturbation of the initial velocity field -- needs nothing but modglobal and decomp_2d.  This script wraps that routine,
# AT BUILD TIME and only into oracle/_ref/, in a module of its own, so that ref_driver.f90's cold start perturbs the
# fields with the reference's compiled code instead of a restatement.  Nothing is edited inside the range and no reference
# text is stored in the repository.
set -e
SRC=${1:?path to the reference src/modstartup.f90}
OUT=${2:?output file}
n=$(wc -l < "$SRC")
[ "$n" -eq 2398 ] || { echo "extract_startup.sh: $SRC has $n lines, expected 2398 (the line range is pinned to this snapshot)" >&2; exit 1; }
{
  echo "module modstartup_rand"
  echo "  implicit none"
  echo "contains"
  sed -n '2367,2396p' "$SRC"
  echo "end module modstartup_rand"
} > "$OUT"
grep -q "subroutine randomize_field" "$OUT" && grep -q "end subroutine randomize_field" "$OUT"
