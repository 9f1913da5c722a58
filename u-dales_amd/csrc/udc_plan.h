// The order of one fused RK3 substep as a pure function of what it depends on (no HIP, no handle): substep_fused (udc_api.hip)
// asks plan_substep() and then only executes.  Host-only C++, also compiled by g++ into lib/libudcplan.so, which the CPU test
// tests/test_substep_plan.py enumerates against the table of DESIGN.md section 7.
#pragma once

struct PlanIn {
  // switches (Switches, udc_internal.h)
  int no_fold, no_alias, ek_always, halo_overlap, mom_pipe, div_in_fft, ptotal;
  // what the handle is
  int slab;             // distributed layout in use (more than one rank, or UDC_FORCE_SLAB)
  int comm_stream;      // the communication stream exists (slab layout set up)
  int sgs;              // 0 DNS, 1 Smagorinsky, 2 Vreman, 3 one-equation
  int lbuoycorr;
  int nslots;           // transported scalars (thl, qt, tke, passive)
  int ibm_on, stats_any;
  int fft_fused;        // own line transforms on the slab path
  int own_fwd;          // own forward half on the single-slab path
  int tend_plane;       // something between the sweep and the solve sums the tendencies where a pressure gradient does not cancel as it
                        // does over the whole periodic volume: masscorr's outflow-rate branch (up(ie, :, :), luoutflowr), or masscorr's
                        // volume flow over the fluid cells only (immersed boundary)
  int between;          // something acts on the tendencies between the momentum sweep and the solve besides the floor: Coriolis,
                        // level forcings, prescribed flow rates, immersed boundary, shifted boundaries, buoyancy / radiative source
  int closure_tile_rows, mom_tile_rows, int_tile_rows;      // tile rows of the three sweeps on this slab
  int x_row_groups;     // row groups of the x forward transform
  int levels_per_chunk; // nz / k-chunks of the transposes
  int p_transpose;      // switch UDC_P_TRANSPOSE: p's ghost rows ride in the backward transpose of the slab solve (own line transforms)
  int open_lid;         // BCtopm = 3 (BCtopm_pressure): w(ke+1) is prognostic -- bcpup, tderive and tstep_integrate have a row there,
                        // taken by three plane kernels beside the sweeps (k_lid_*, udc_pois.hip)
  int lid_masked;       // ... and obstacles reach level ke: the lid's slab means run over the fluid c cells only (avexy_ibm)
  // this call
  int rk3step;
  int um_alias;         // um, vm, wm are logically u0, v0, w0 (the previous substep was an aliased stage 3)
  int ibm_edits_now;    // this substep runs ibmwallfun / ibmnorm
};

enum PlanClosure { CLOSURE_FOLDED = 0, CLOSURE_OVERLAPPED = 1, CLOSURE_PLAIN = 2 };
enum PlanRow { ROW_FOLDED = 0, ROW_BESIDE = 1, ROW_INLINE = 2, ROW_PIPED = 3, ROW_TRANSPOSED = 4 };
enum PlanIntegrate { INT_ONE = 0, INT_EDGES_FIRST = 1 };

struct Plan {
  int lds, pup, fold, alias_ok;
  int materialise_um;   // um is made a real copy again before anything else
  int rotate;           // stage 1 after an aliased stage 3: reads u0 as um, writes into the stale um buffers, swaps the pointers
  int skip_um;          // stage 3 leaves um unwritten (aliased from now on)
  int closure;          // PlanClosure
  int need_ekh;         // the closure writes (and, on y-slabs, exchanges) ekh as well: always with the plain kernel
  int mom_pipe;         // momentum sweep cut along the solve's k-chunks, under the forward transposes
  int div_in_fft;       // fillps' divergence inside the x forward transform
  int vp_row;           // PlanRow: vp's ghost row
  int p_row;            // PlanRow: p's ghost row (FOLDED / inside the backward TRANSPOSE / BESIDE the first interior rows / INLINE)
  int integrate;        // PlanIntegrate
  int ptotal;           // pressure-total form: the momentum sweep leaves the gradient of pres0 out, the solve returns pres0 + p, the
                        // projection applies it as a whole and it becomes pres0 (arrays swapped): pres0 is read nowhere in the substep
};

inline bool plan_halo_overlap(const PlanIn &in, int tile_rows) { return in.slab && in.comm_stream && in.halo_overlap && tile_rows >= 3; }

inline Plan plan_substep(const PlanIn &in) {
  Plan p{};
  p.lds = 1;      // (the LDS-staged sweeps and the predicted-velocity form of the tendencies: always, since round 4)
  p.pup = 1;
  // single slab (whole y extent local): the ghost rows / planes of closurebc, bcpup, bcp, halos and boundary are written by the
  // kernels that own the neighbouring cells
  p.fold = p.lds && !in.slab && !in.no_fold;
  // um aliasing: stage 3 leaves um unwritten (== u0); stage 1 reads u0 in its place (ibmnorm edits um: not with obstacles)
  // (open lid: wm(ke+1) is read and written by the lid's plane kernels under its own name)
  p.alias_ok = p.pup && !in.no_alias && !in.ibm_on && !in.open_lid;
  p.materialise_um = in.um_alias && !(p.alias_ok && in.rk3step == 1);
  p.rotate = in.um_alias && !p.materialise_um;
  p.skip_um = p.alias_ok && in.rk3step == 3;
  const bool smag_vreman = in.sgs == 1 || in.sgs == 2;
  if (p.fold && smag_vreman && !in.lbuoycorr) {
    p.closure = CLOSURE_FOLDED;
    // ekh has readers only where a scalar is transported, or between time steps (maxima, statistics, restart files: RK stage 3)
    p.need_ekh = in.ek_always || in.rk3step == 3 || in.nslots > 0 || in.stats_any;
  } else if (p.lds && smag_vreman && !in.lbuoycorr && plan_halo_overlap(in, in.closure_tile_rows)) {
    p.closure = CLOSURE_OVERLAPPED;
    p.need_ekh = in.ek_always || in.rk3step == 3 || in.nslots > 0 || in.stats_any;      // (as folded: one array and one ghost row less)
  } else {
    p.closure = CLOSURE_PLAIN;
    p.need_ekh = 1;
  }
  p.mom_pipe = in.slab && p.lds && p.pup && in.mom_pipe && in.fft_fused && in.div_in_fft && plan_halo_overlap(in, in.mom_tile_rows) &&
               in.nslots == 0 && in.sgs != 3 && !in.between && in.x_row_groups >= 2 && in.levels_per_chunk >= 4 && !in.open_lid;
  // (open lid: the divergence of level ke reads pwp(ke+1), the plane bcpup's lid kernel fills ahead of the solve: the transform reads it too)
  p.div_in_fft = p.pup && ((in.slab && in.fft_fused && in.div_in_fft) || (!in.slab && in.own_fwd));
  if (p.mom_pipe) p.vp_row = ROW_PIPED;
  else if (!p.fold || (in.ibm_on && in.ibm_edits_now))
    p.vp_row = (p.div_in_fft && plan_halo_overlap(in, 3) && in.x_row_groups >= 2) ? ROW_BESIDE : ROW_INLINE;
  else p.vp_row = ROW_FOLDED;
  if (p.fold) { p.p_row = ROW_FOLDED; p.integrate = INT_ONE; }
  else {
    // y-slabs with the own line transforms: the y pass leaves every slab's edge rows in the neighbours' blocks as well and the x pass
    // transforms them with the rest (2 rows in ny_l more per transpose) -- no exchange of p's rows, no sweep to hide it behind
    if (in.slab && in.fft_fused && in.p_transpose) p.p_row = ROW_TRANSPOSED;
    else p.p_row = (plan_halo_overlap(in, in.int_tile_rows) && in.int_tile_rows >= 4) ? ROW_BESIDE : ROW_INLINE;
    p.integrate = plan_halo_overlap(in, in.int_tile_rows) ? INT_EDGES_FIRST : INT_ONE;
  }
  // The reference adds -grad pres0 to the tendencies (advecu/v/w) and solves for the increment p (fillps .. tderive, pres0 += p).  The
  // discrete operators are the same on both sides (the solver's matrix IS div grad, floor and lid rows included), so solving for
  // pres0 + p from tendencies without the old gradient gives the same velocities and the same pres0 up to round-off -- and 24 B per
  // cell less (pres0 read by the sweep, read and written by the projection).  ibmnorm's zero tendency at a solid point includes the
  // old gradient: there the tendency becomes + that gradient instead (k_ibm_norm).  Where something sums the tendencies over less than
  // the periodic volume (the outflow-rate mass correction; the volume flow over the fluid cells of an immersed boundary) the two
  // forms differ: not there.  On y-slabs p's ghost row then travels both ways (it is pres0's), and pres0 leaves the
  // exchange of the new velocities' rows.
  // Open lid (last session of round 6): the matrix is the closed lid's plus, on the zero mode only, the Dirichlet row "p = 0 on the top face"
  // (src/modpois.f90:207-217), i.e. A = L + T with (T q)(ke) = -2 <q>(ke) dzhi(ke+1) dzfi(ke).  The reference solves A p = div(pup_ref) with
  // bcpup's lid row pwp(ke+1) = wm(ke+1) / rk3coef + 2 <pres0>(ke) dzhi(ke+1); adding A pres0 = L pres0 + T pres0 on both sides, the sum
  // pres0 + p solves A (pres0 + p) = div(pup without grad pres0) with the lid row WITHOUT its pres0 term -- T pres0 cancels it exactly.  So
  // the form holds with lid_bcpup_kernel leaving that term out (and, unlike under a closed lid, the constant of pres0 + p is the
  // reference's too: the zero mode is not singular).  tderive's row 2 <p>(ke) dzhi(ke+1) then takes the mean of the sum.  Not where
  // obstacles reach the lid: avexy_ibm's mean over the fluid cells of level ke is not the zero mode's.
  p.ptotal = in.ptotal && p.pup && !in.tend_plane && !(in.open_lid && in.lid_masked);
  return p;
}
