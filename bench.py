#!/usr/bin/env python3
"""bench.py -- cell-updates/s of the uDALES dynamical core (advect + diffuse + Poisson + RK3) on MI355X.

A "step" is ONE RK3 substep of the hot path (src/program.f90:132-222 restricted to advection,
subgrid, forces, poisson, tstep_integrate, halos, boundary) over the whole grid, i.e. one
cell-update per cell.  Workload at N=1: BASELINE.json configs[1], the
256^3 neutral empty-domain channel (2nd-order advection, Vreman SGS = the reference default,
FFT Poisson), synthetic cold start (LCG noise of src/modstartup.f90:2367-2396), fields resident
in HBM before the timed region.  Workload at N>1: BASELINE.json configs[3], the 1024x512x512 channel
split into N y-slabs (STRONG scaling: the same grid at every N; --size 1024x512x512 gives the N=1 point,
and every N>1 line also carries rank 0's own one-GPU run of that grid, `single_gpu_same_workload`);
--weak switches to one 256^3 slab per GPU.  `poisson_only_ms` times the reference's `poisson` routine
alone (fillps + solve + tderive), the subject of the >= 6x strong-scaling target.

    python bench.py                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed live with HIP events
recorded on the library's own stream during the timed region; `cpu_baseline` times the reference's
own Fortran (oracle/_ref/udales_ref, see oracle/Makefile) on a bounded sample on this box's host.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "u-dales_amd"))

# Algorithmic (compulsory) bytes per cell-update of each kernel AS BUILT.
# SURVEY.md section 8d's model is closure 40 + momentum 88 + Poisson/integrate 264 = 392 B; the fused
# substep moves less than that model (no pup/pvp/pwp or rhs arrays, tendencies neither re-read nor
# zero-filled): 40 + 88 + 32 + 64 + 20 + 72 = 316 B.  `model_392B_ratio` keeps SURVEY's 392 B
# definition so that it stays comparable across rounds.
SCALAR_INTEGRATE_BYTES = 24      # per transported scalar in project_integrate: read svp, svm; write sv0
ALGO_BYTES = {
    "closure": 40,              # read u0,v0,w0; write ekm,ekh -- ekh only where something reads it: with no transported scalar and no
                                # statistics that is RK stage 3 alone (the time-step limit of tstep_update): 32 + 8/3, charged below
    "mom": 88,                  # read u0,v0,w0,pres0,ekm (40) + um,vm,wm (24); write pup,pvp,pwp (24); on RK stage 1 um is u0
                                # (buffer rotation, already staged): 64 -- the timed launches are charged their own mix
    "div_rhs": 32,              # read pup,pvp,pwp; write p
    "fft_fwd": 32, "fft_bwd": 32,   # 2 passes x (8 read + 8 write)
    "thomas": 20,               # x read once, written once (16) + the pivot table read once (4) -- as built since round 5 (SURVEY 8d's
                                # model has a second table, 24 B; the back substitution's coefficient is now formed in the kernel);
                                # one GPU: the mirrored rows ky, ny - ky share the table, 18 B
    "project_integrate": 72,    # read p (8), pup,pvp,pwp (24); RMW pres0 (16); write u0,v0,w0 (24)
    "scalar": 48,               # read c, ekh, u0,v0,w0 (40); write cp (8) -- tendencies are not re-read in the fused substep
    # slab (multi-GPU) Poisson stages; on one GPU the forward half carries the first two names too (udc_fft.hip: divergence + x transform
    # into rocFFT's spectral layout = "fftx_pack_fwd" at 32 B, the y pass over it = "unpack_ffty_fwd" at 16 B; 40 + 88 + 32 + 16 + 20 + 32 + 72 = 300 B)
    # (own line FFTs reading / writing the exchange buffers directly: one real field in, one out per stage, DESIGN.md section 6;
    #  the x forward stage of the fused substep evaluates the divergence itself -- reads pup, pvp, pwp instead of p: 24 + 8 = 32,
    #  charged below when the substep launched no div_rhs)
    "fftx_pack_fwd": 16, "unpack_ffty_fwd": 16, "ffty_pack_bwd": 16, "unpack_fftx_bwd": 16,
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0    # ... and what the same guide gives as achievable


MOM_STAGE1_SAVING = 24         # um, vm, wm are not read on RK stage 1 of the fused substep (um aliases u0)


def fold_edges(tab):
    """y-slabs: a kernel whose output rows travel to the neighbours is launched twice (`<name>_edge` over the tile rows next to the
    neighbouring ranks, then `<name>` over the rest, the exchange in between on the communication stream): one entry, the launches
    of the second."""
    out = dict(tab)
    for k in [k for k in out if k.endswith("_edge")]:
        ms, cnt = out.pop(k)
        base = k[:-5]
        if base in out:
            out[base] = (out[base][0] + ms, out[base][1])
        else:
            out[base] = (ms, cnt)
    return out


SCALARS_INLINE = 0              # set from the executed plan: scalars whose RK3 update rode in their sweep
PRESSURE_TOTAL = False          # set from the executed plan: the substep ran in the pressure-total form (pres0 read nowhere: the momentum
                                # sweep 8 B less, the projection 16 B less per cell)


def algo_bytes(name, nscal=0, stage1_frac=0.0):
    for k, v in ALGO_BYTES.items():
        if name.startswith(k):
            if k == "mom":
                return v - MOM_STAGE1_SAVING * stage1_frac - (8 if PRESSURE_TOTAL else 0)
            if k == "project_integrate":
                return v + SCALAR_INTEGRATE_BYTES * (nscal - SCALARS_INLINE) - (16 if PRESSURE_TOTAL else 0)
            if k == "scalar" and SCALARS_INLINE:
                # (svm read too, the new value written where the tendency would go: 56 B; the integration skips the scalar) -- for the
                # scalars that took the update in their sweep only: the row averages over all scalar launches
                return v + 8.0 * min(SCALARS_INLINE, max(nscal, 1)) / max(nscal, 1)
            if k == "closure" and nscal == 0 and os.environ.get("UDC_EK_ALWAYS", "0") in ("", "0"):
                return v - 8 * 2.0 / 3.0
            return v + (SCALAR_INTEGRATE_BYTES * nscal if k == "project_integrate" else 0)
    return None


def cube_array_ibm(core, nx, ny, nz, j0=0, edge=32, pitch=128, iwallmom=3, heat=False):
    """A staggered array of cubes (edge cells every pitch cells, the layout of BASELINE configs[4]) handed to the core: solid and
    fluid-boundary point lists of the four grids and one facet section per boundary point for the wall functions (an oblique
    facet normal, so that nothing is skipped and the stress is rotated; every third section takes the velocity at a
    reconstruction point).  Global lists: every slab keeps its rows.  -> counts."""
    import numpy as np
    g = core.g
    c = np.zeros((nz + 2, ny, nx), dtype=bool)
    for jb, jj in enumerate(range(pitch // 4, ny - edge, pitch)):
        for ii in range(pitch // 4 + (pitch // 2 if jb % 2 else 0), nx - edge, pitch):
            c[1:edge + 1, jj:jj + edge, ii:ii + edge] = True
    u = c | np.roll(c, 1, axis=2)
    v = c | np.roll(c, 1, axis=1)
    w = c.copy(); w[1:] |= c[:-1]
    lists = {}
    for name, sol in (("u", u), ("v", v), ("w", w), ("c", c)):
        nb = np.zeros_like(sol)
        for ax, sh in ((2, 1), (2, -1), (1, 1), (1, -1)):
            nb |= np.roll(sol, sh, axis=ax)
        nb[1:] |= sol[:-1]; nb[:-1] |= sol[1:]
        bnd = nb & ~sol

        def pts(m, lo):
            m = m.copy(); m[:lo] = False; m[nz + 1:] = False
            kji = np.argwhere(m)
            return np.ascontiguousarray(np.stack([kji[:, 2] + 1, kji[:, 1] + 1, kji[:, 0]], axis=1), dtype=np.int32)
        lists[name] = (pts(sol, 1), pts(bnd, 2 if name == "w" else 1))
    for q, name in enumerate("uvwc"):
        if name == "c" and not (core.nsv or getattr(core, "ltempeq", False)):
            continue
        core.set_ibm_points(q, *lists[name])
    core.ibm_commit()
    dx, dy = g.dx, g.dy
    xh, xf = np.arange(nx + 1) * dx, (np.arange(nx + 1) + 0.5) * dx
    yh, yf = np.arange(ny + 1) * dy, (np.arange(ny + 1) + 0.5) * dy
    zf, zh = g.zf[1:nz + 2], g.zh[1:nz + 2]
    facets = {"norm": np.array([[0.6, 0., 0.8]]), "z0": np.array([0.01]), "z0h": np.array([0.001]), "tsurf": np.array([289.])}
    core.set_ibm_wallfun(iwallmom, 0.71, zf, zh)
    counts = {"solid_c_cells": int(c.sum()), "sections": {}}
    for q, name in enumerate("uvwc" if heat else "uvw"):
        bnd = lists[name][1]
        n = len(bnd)
        xg, yg, zg = {"u": (xh, yf, zf), "v": (xf, yh, zf), "w": (xf, yf, zh), "c": (xf, yf, zf)}[name]
        i, j, k = bnd[:, 0], bnd[:, 1], bnd[:, 2]
        p = np.stack([xg[i - 1] + 0.3 * dx, yg[j - 1] + 0.2 * dy, zg[k - 1] + 0.3 * (zf[1] - zf[0])], axis=1)
        ids = np.zeros((n, 4, 3), dtype=np.int32)
        for t, (a, b, cc) in enumerate(((xh, yf, zf), (xf, yh, zf), (xf, yf, zh), (xf, yf, zf))):
            ids[:, t, 0] = np.searchsorted(a, p[:, 0], side="right")
            ids[:, t, 1] = np.searchsorted(b, p[:, 1], side="right")
            ids[:, t, 2] = np.searchsorted(cc, p[:, 2], side="right")
        ok = (ids.min(axis=(1, 2)) >= 1) & (ids[:, :, 0].max(axis=1) <= nx) & (ids[:, :, 1].max(axis=1) <= ny) & (ids[:, :, 2].max(axis=1) <= nz)
        rec = ok & (np.arange(n) % 3 == 0)
        S = {"n": n, "cell": bnd, "area": np.full(n, 0.1), "dist": np.full(n, 0.25), "fac": np.ones(n, dtype=np.int32),
             "comprec": np.where(rec, 0, 1).astype(np.int32), "recpt": p, "recids": np.where(rec[:, None, None], ids, 1).astype(np.int32)}
        core.set_ibm_sections(q, S, facets, np.ones((n, 2)))
        counts["sections"][name] = {"sections": int(n), "with_reconstruction": int(rec.sum())}
    if heat:
        core.set_ibm_wallheat(2)
    counts["points"] = {n: {"solid": int(len(lists[n][0])), "boundary": int(len(lists[n][1]))} for n in "uvwc"}
    return counts


def workload_name(nx, ny, nz, sgs, nsv, floor, ibm=False):
    """What actually runs, and the BASELINE.json configuration it is -- only when it is one."""
    w = (f"{nx}x{ny}x{nz} neutral empty-domain channel, cd2 momentum advection + "
         f"{'Vreman' if sgs == 'vreman' else 'Smagorinsky'} SGS diffusion"
         f"{' + ' + str(nsv) + ' kappa-advected scalar' + ('s' if nsv > 1 else '') if nsv else ''}"
         f" + FFT(x,y)/tridiagonal(z) Poisson + RK3 substep")
    if (nx, ny, nz, sgs, nsv) == (256, 256, 256, "vreman", 0):
        w += " (BASELINE.json configs[1])"
    elif (nx, ny, nz, sgs, nsv) == (512, 512, 256, "smag", 1):
        w += " (BASELINE.json configs[2])"
    elif (nx, ny, nz, sgs, nsv) == (1024, 512, 512, "vreman", 0):
        w += " (BASELINE.json configs[3])"
    if ibm:
        w = w.replace("neutral empty-domain channel", "channel over a staggered array of 32-cell cubes every 128 cells (immersed boundary: "
                      "masked stencils, sparse corrections, neutral facet wall functions on every boundary point)")
        w = w.replace(" (BASELINE.json configs[1])", "").replace(" (BASELINE.json configs[2])", "").replace(" (BASELINE.json configs[3])", "")
        if (nx, ny, nz) == (512, 512, 512):
            w += " (the grid and obstacle layout of BASELINE.json configs[4])"
    return w


def time_loop(core, fn, n, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    barrier()
    return (time.perf_counter() - t0) / n * 1e3


def write_deck(d, iexp, nx, ny, nz, nsub, dt=0.25, nprocy=1, nsv=0, sgs="vreman", floor=True, nwarm=0, runtime=1000000.):
    with open(os.path.join(d, f"namoptions.{iexp:03d}"), "w") as f:
        f.write(f"""&RUN
iexpnr = {iexp}
runtime = {runtime!r}
dtmax = {dt}
ladaptive = .false.
irandom = 43
randu = 0.01
nprocx = 1
nprocy = {nprocy}
libm = .false.
/
&DOMAIN
itot = {nx}
jtot = {ny}
ktot = {nz}
xlen = {nx * 0.5}
ylen = {ny * 0.5}
/
&PHYSICS
/
&DYNAMICS
ipoiss = 0
/
&BC
{('BCbotm = 3' + chr(10) + 'z0 = 0.05') if floor else ''}
/
{('&WALLS' + chr(10) + 'nfcts = 0' + chr(10) + 'lbottom = .true.' + chr(10) + '/') if floor else ''}
&SCALARS
nsv = {nsv}
/
&NAMSUBGRID
{'lvreman = .true.' if sgs == 'vreman' else 'lsmagorinsky = .true.' + chr(10) + 'lvreman = .false.'}
/
&ORACLE
nsub = {nsub}
nwarm = {nwarm}
/
""")
    with open(os.path.join(d, f"prof.inp.{iexp:03d}"), "w") as f:
        f.write("# bench\n# z thl qt u v tke\n")
        for k in range(nz):
            f.write(f"{(k + 0.5) * 0.5:.15f} 288.0 0.0 1.0 0.0 0.0\n")
    with open(os.path.join(d, f"lscale.inp.{iexp:03d}"), "w") as f:
        f.write("# bench\n# z uq vq pqx pqy wfls dqtdxls dqtdyls dqtdtls dthlrad\n")
        for k in range(nz):
            f.write(f"{(k + 0.5) * 0.5:.15f} 0.0 0.0 0.0001 0.0 0.0 0.0 0.0 0.0 0.0\n")
    if nsv > 0:      # scalar.inp (src/modstartup.f90:1541-1548): scalar n = n z / zsize, so that the kappa limiter has work
        with open(os.path.join(d, f"scalar.inp.{iexp:03d}"), "w") as f:
            f.write("# bench\n# z sv(1..nsv)\n")
            for k in range(nz):
                f.write(f"{(k + 0.5) * 0.5:.15f} " + " ".join(repr((n + 1) * (k + 0.5) / nz) for n in range(nsv)) + "\n")
    return os.path.join(d, f"namoptions.{iexp:03d}")


def _run_ref(cmd, cwd, env=None, raw=False):
    try:
        r = subprocess.run(f"ulimit -s unlimited; exec {cmd}", shell=True, cwd=cwd, capture_output=True,
                           text=True, timeout=900, executable="/bin/bash", env=env)
    except subprocess.TimeoutExpired:
        return None
    m = re.search(r"cell_updates_per_s=\s*([0-9.Ee+-]+)", r.stdout)
    if raw:
        return (float(m.group(1)) if m else None), r.stdout
    return float(m.group(1)) if m else None


def dropin_leg(nx, ny, nz, nsv, sgs, floor, value, nsub=450, nwarm=30):
    """The same workload through the drop-in boundary: the reference-shaped Fortran driver (call order of
    src/program.f90:132-222) linked with the drop-in modules of u-dales_amd/fortran/ over the C ABI, device resident
    (UDC_RESIDENCY=2).  Wall clock around the Fortran time loop (MPI_Wtime + a final device synchronisation)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "udales_dropin")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        write_deck(tmp, 902, nx, ny, nz, nsub, nsv=nsv, sgs=sgs, floor=floor, nwarm=nwarm)
        env = dict(os.environ, UDC_RESIDENCY="2")
        res = _run_ref(f"{exe} namoptions.902 time none.bin", tmp, env=env, raw=True)
    if not res or not res[0]:
        return {"value": None, "note": "udales_dropin failed"}
    v, out = res
    m = re.search(r"fused_substeps=(\d+) unfused=(\d+)", out)
    dm = re.search(r"divmax=\s*([0-9.Ee+-]+)", out)
    # ... and through the reference's OWN main program (u-dales_amd/bin/udales_full_dropin: src/program.f90 untouched, every other file of
    # the reference's src/ but the ten drop-in modules), timed by the reference's own clock around its loop (`TOTAL CPU time by
    # main time loop`, src/modmpi.f90:140-160) -- which, unlike the figure above, includes the first substep's upload of the state
    # and the last one's download for the restart / output code
    real = None
    exe2 = os.path.join(ROOT, "u-dales_amd", "bin", "udales_full_dropin")
    if os.path.exists(exe2):
        nstep = 400
        with tempfile.TemporaryDirectory() as tmp:
            write_deck(tmp, 904, nx, ny, nz, 0, nsv=nsv, sgs=sgs, floor=floor, runtime=nstep * 0.25 - 1e-6)
            try:
                r2 = subprocess.run(f"ulimit -s unlimited; exec {exe2} namoptions.904", shell=True, cwd=tmp, capture_output=True, text=True,
                                    timeout=900, executable="/bin/bash", env=dict(os.environ, UDC_RESIDENCY="2", UDC_TIMERS="1"))
                mt = re.search(r"TOTAL CPU time by main time loop =\s*([0-9.Ee+-]+)", r2.stdout)
                if r2.returncode == 0 and mt:
                    sec = float(mt.group(1))
                    # the drop-in modules' host-side phase clock (UDC_TIMERS=1): the one-time upload of the state when the loop starts and
                    # its download for the output / restart code when it ends, and what the host spends inside the drop-in routines
                    phases = {m_.group(1): {"calls": int(m_.group(2)), "seconds": float(m_.group(3))}
                              for m_ in re.finditer(r"UDC_TIMER\s+(\w+)\s+calls=\s*(\d+)\s+seconds=\s*([0-9.Ee+-]+)", r2.stdout)}
                    once = sum(phases.get(k, {}).get("seconds", 0.) for k in ("upload", "download"))
                    steady = max(sec - once, 1e-9)
                    real = {"value": nx * ny * nz * 3 * nstep / sec, "unit": "cell-updates/s", "steps": nstep, "loop_seconds": round(sec, 4),
                            "frac_of_direct": round(nx * ny * nz * 3 * nstep / sec / value, 4),
                            "host_phases": phases,
                            "one_time_transfers_s": round(once, 4),
                            "steady_state": {"value": nx * ny * nz * 3 * nstep / steady, "ms_per_substep": round(steady / (3 * nstep) * 1e3, 5),
                                             "frac_of_direct": round(nx * ny * nz * 3 * nstep / steady / value, 4),
                                             "note": "the loop's time minus the one-time upload / download of the state"},
                            "surface": "u-dales_amd/bin/udales_full_dropin namoptions.NNN: the reference's program.f90, modstartup.f90 and every "
                                       "other file of its src/ unmodified, minus the ten drop-in modules; its own timer around its loop "
                                       "(incl. the one-time upload / final download of the state)"}
            except subprocess.TimeoutExpired:
                real = None
    return {"value": v, "reference_main_program": real, "unit": "cell-updates/s", "frac_of_direct": round(v / value, 4), "substeps": nsub, "warmup": nwarm,
            "ms_per_step": round(nx * ny * nz / v * 1e3, 5),
            "fused_substeps": int(m.group(1)) if m else None, "unfused_substeps": int(m.group(2)) if m else None,
            "divmax_after_run": float(dm.group(1)) if dm else None,
            "surface": "the reference's whole src/ tree minus the ten drop-in modules (modstartup's start-up, every call of "
                       "src/program.f90:132-222 per substep incl. checksim, statsdump, thermodynamics; main program oracle/ref_driver.f90 "
                       "for the timer around the loop) -> drop-in modules -> C ABI, UDC_RESIDENCY=2"}


def cpu_baseline(nx, ny, nz, budget_s=25.0):
    """Reference CPU path (the reference's own Fortran, oracle/_ref) on a bounded number of substeps:
    single rank, and -- when MPICH is present -- one MPI rank per core over a y-slab decomposition."""
    ref = os.path.join(ROOT, "oracle", "_ref", "udales_ref")
    ref_mpi = os.path.join(ROOT, "oracle", "_ref", "udales_ref_mpi")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(ref):
        return None
    cells = nx * ny * nz
    nsub = max(1, min(6, int(budget_s * 3.0e6 / cells)))     # ~3e6 cell-updates/s/core expected
    caveat = ("reference Fortran (flang -O3, real(8)) over repo-owned stand-ins for the absent "
              "2decomp-fft/FFTW layers (real line transforms as half-length complex FFT + split, no allocation per line: "
              "FFT_REF_FAST=1), wall clock as src/modmpi.f90:140-160")
    fenv = dict(os.environ, FFT_REF_FAST="1")
    with tempfile.TemporaryDirectory() as tmp:
        write_deck(tmp, 900, nx, ny, nz, nsub)
        v1 = _run_ref(f"{ref} namoptions.900 time none.bin", tmp, env=fenv)
        out = None
        if v1:
            out = {"value": v1, "unit": "cell-updates/s", "cores": 1, "kind": "reference",
                   "sample": f"{nsub} RK3 substeps of the same {nx}x{ny}x{nz} channel, single rank; {caveat}"}
        if os.path.exists(ref_mpi) and os.path.exists(mpiexec):
            ncpu = os.cpu_count() or 1
            scan = {}
            for p in (8, 16, 32, 64):
                if p > max(ncpu // 2, 1) or ny % p or nz % p or ny // p < 4:
                    continue
                write_deck(tmp, 900, nx, ny, nz, 2 * nsub, nprocy=p)
                vp = _run_ref(f"{mpiexec} -n {p} {ref_mpi} namoptions.900 time none.bin", tmp, env=fenv)
                if vp:
                    scan[p] = vp
            if scan:
                p = max(scan, key=scan.get)
                # the best rank count once more on a sample of >= 30 substeps (bounded: ~10 s at the scan's rate)
                nlong = max(30, 2 * nsub)
                if scan[p] * 12.0 < cells * nlong:
                    nlong = max(2 * nsub, int(scan[p] * 12.0 / cells))
                vlong = None
                if nlong > 2 * nsub:
                    write_deck(tmp, 900, nx, ny, nz, nlong, nprocy=p)
                    vlong = _run_ref(f"{mpiexec} -n {p} {ref_mpi} namoptions.900 time none.bin", tmp, env=fenv)
                best = vlong if vlong else scan[p]
                if out is None or best > v1:
                    out = {"value": best, "unit": "cell-updates/s", "cores": p, "cores_present": ncpu, "kind": "reference",
                           "single_core_value": v1, "ranks_scan": scan,
                           "sample": f"{nlong if vlong else 2 * nsub} RK3 substeps of the same {nx}x{ny}x{nz} channel on {p} MPI ranks of the "
                                     f"{ncpu} hardware threads present (best of a scan over {sorted(scan)} ranks on {2 * nsub} substeps each; MPICH, "
                                     f"nprocx=1, nprocy=P: y-slabs, z<->y transposes as pairwise MPI_SENDRECV rounds); {caveat}"}
    return out



INVARIANCE_TOL = 1.0e-9      # ABS_TOL of the reference's decomposition test (tests/integration/processor_boundaries/
                             # test_processor_boundaries.py:28) on fields of order one; here relative to each field's maximum


def single_gpu_leg(nx, ny, nz, dt, args, device, ms_n, poisson_n, poisson_sub_n, init=None, after=None, nsub_inv=0):
    """One GPU, the same grid, the single-slab code path.  With `init` (global interior u0, v0, w0 of the run's cold start, torch
    tensors on this device or numpy) the leg starts from the SAME state as the slab ranks did, runs `nsub_inv` substeps and
    compares u0, v0, w0, pres0 with `after` (the slabs' fields after the same substeps, gathered): the decomposition-invariance
    record of the line.  Then timing, from wherever the state is."""
    import numpy as np
    import torch
    import udcore
    from udcore import read_deck
    with tempfile.TemporaryDirectory() as tmp:
        deck1 = read_deck(write_deck(tmp, 903, nx, ny, nz, 0, dt=dt, nprocy=1, nsv=args.nsv, sgs=args.sgs, floor=not args.no_floor))
    c1 = udcore.from_deck(deck1, device=device, rank=0, nranks=1)
    if args.ibm:
        cube_array_ibm(c1, nx, ny, nz)
    inv = None
    if init is not None:
        for k in ("u0", "v0", "w0"):
            a = init[k].cpu().numpy() if hasattr(init[k], "cpu") else init[k]
            c1.upload(k, a)
            c1.upload(k.replace("0", "m"), a)
            del a
    else:
        rng = np.random.default_rng(43)
        noise = 0.02 * (rng.random(c1.g.mshape()) - 0.5)
        for k, base in (("u0", 1.0), ("v0", 0.0), ("w0", 0.0)):
            a = noise + base
            c1.upload(k, a)
            c1.upload(k.replace("0", "m"), a)
        del noise, a
    c1.halos()
    c1.boundary()

    def sync1():
        torch.cuda.synchronize()
        c1.sync()
    rk1 = [1]

    def step1():
        c1.substep(rk1[0], dt, True)
        rk1[0] = rk1[0] % 3 + 1
    if init is not None and after is not None and nsub_inv > 0:
        for _ in range(nsub_inv):
            step1()
        sync1()
        worst = {}
        for k in ("u0", "v0", "w0", "pres0"):
            ref = torch.from_numpy(c1.download(k)[1:-1, 1:-1, 1:-1])
            got = after[k]
            got = got if hasattr(got, "cpu") else torch.from_numpy(got)
            ref = ref.to(got.device)
            scale = max(float(ref.abs().max()), 1e-30)
            worst[k] = float((got - ref).abs().max()) / scale
            del ref, got
        inv = {"substeps": nsub_inv, "max_rel_diff": {k: float(f"{v:.3e}") for k, v in worst.items()}, "tolerance": INVARIANCE_TOL,
               "ok": bool(all(v <= INVARIANCE_TOL for v in worst.values())),
               "against": "rank 0's one-GPU run (single-slab code path) of the same grid from the same cold start, u0 v0 w0 pres0 "
                          "of every cell, max |slabs - one GPU| / max |one GPU|"}
    for _ in range(6):
        step1()
    n1 = max(6, min(args.steps, 30))
    c1.profile(True)
    c1.profile_reset()
    ms1 = time_loop(c1, step1, n1, sync1)
    prof1 = c1.profile_get()
    c1.profile(False)
    nonpois = sum(ms for name, (ms, cnt) in prof1.items() if name.startswith(("closure", "mom_", "bottom", "scalar"))) / n1
    c1.rk3step, c1.dt = 1, dt
    for _ in range(3):
        c1.poisson()
    p1 = time_loop(c1, c1.poisson, 10, sync1)
    c1.close()
    out = {"ms_per_step": round(ms1, 5), "poisson_only_ms": round(p1, 5), "poisson_in_substep_ms": round(ms1 - nonpois, 5),
           "steps": n1, "speedup_substep": round(ms1 / ms_n, 3), "speedup_poisson": round(p1 / poisson_n, 3),
           "speedup_poisson_in_substep": round((ms1 - nonpois) / poisson_sub_n, 3)}
    return out, inv


# ---------------------------------------------------------------------------------------------------------------------------------
# N > 1: every rank that torch.distributed.run starts is a SUPERVISOR.  It never touches a GPU; it starts the real bench (this file
# with --worker) as a child with its own rendezvous port, watches the child's heartbeat file, and agrees with the other supervisors
# (gloo over the launcher's own MASTER_ADDR / MASTER_PORT, one small all-reduce every half second) on what happened.  A child that
# dies, or whose heartbeat stands still for --stall-timeout seconds (a hung collective), fails the attempt on EVERY rank: all
# children of the attempt are killed and the next rung of the ladder is tried -- the same workload, the same timed region, with the
# slab substep's overlap features taken back one group at a time.  The line says which rung produced the number.
LADDER = [
    ("defaults", {}),
    ("ghost rows (p's among them, in an exchange of their own) and sweeps in line", {"UDC_HALO_OVERLAP": "0", "UDC_MOM_PIPE": "0", "UDC_P_TRANSPOSE": "0"}),
    ("... and the transposes in one piece", {"UDC_HALO_OVERLAP": "0", "UDC_MOM_PIPE": "0", "UDC_P_TRANSPOSE": "0", "UDC_A2A_CHUNKS": "1"}),
    ("... and rocFFT + transpose kernels instead of the fused line transforms",
     {"UDC_HALO_OVERLAP": "0", "UDC_MOM_PIPE": "0", "UDC_P_TRANSPOSE": "0", "UDC_A2A_CHUNKS": "1", "UDC_FFT_FUSED": "0"}),
    ("... and RCCL without peer-to-peer transport (through host memory: degraded links, a number of last resort)",
     {"UDC_HALO_OVERLAP": "0", "UDC_MOM_PIPE": "0", "UDC_P_TRANSPOSE": "0", "UDC_A2A_CHUNKS": "1", "UDC_FFT_FUSED": "0", "NCCL_P2P_DISABLE": "1"}),
]


def heartbeat(phase):
    """Worker side: one line per phase into the file the supervisor watches (no file: not supervised)."""
    fn = os.environ.get("UDC_BENCH_HEARTBEAT")
    if fn:
        with open(fn, "a") as f:
            f.write(f"{time.time():.3f} {phase}\n")
    inj = os.environ.get("UDC_BENCH_INJECT", "")      # tests only: "rung:phase:hang|exit[:rank]" makes that rung fail at that phase
    if inj:
        parts = inj.split(":")
        if (int(parts[0]) == int(os.environ.get("UDC_BENCH_RUNG", "-1")) and parts[1] == phase
                and (len(parts) < 4 or int(parts[3]) == int(os.environ.get("RANK", "0")))):
            if parts[2] == "hang":
                time.sleep(1e6)
            raise SystemExit(7)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def supervise(args, argv):
    import signal
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    tmp = tempfile.mkdtemp(prefix=f"udc_bench_r{rank}_")
    attempts, line, fallback_line = [], None, None
    rungs = LADDER if not args.no_ladder else LADDER[:1]
    for ri, (name, env_extra) in enumerate(rungs):
        port = [free_port() if rank == 0 else 0]
        dist.broadcast_object_list(port, src=0)
        hb, fo, fe = (os.path.join(tmp, f"{x}{ri}") for x in ("hb", "out", "err"))
        # (the launcher's agent store belongs to the supervisors: the children rendezvous by themselves on their own port)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update(MASTER_PORT=str(port[0]), UDC_BENCH_HEARTBEAT=hb, UDC_BENCH_RUNG=str(ri), **env_extra)
        if ri > 0:      # a fallback rung runs the workload and the checks, not the side measurement of the exchanges
            env["UDC_BENCH_NO_EXCHANGE_ACCOUNT"] = "1"
        if env.get("UDC_TEST_SHM"):      # test transport: a segment name per attempt (a killed attempt leaves its segment behind)
            env["UDC_TEST_SHM"] = f"{env['UDC_TEST_SHM']}_{ri}_{port[0]}"
        open(hb, "w").close()
        t0 = time.time()
        with open(fo, "w") as so, open(fe, "w") as se:
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ["--worker"], env=env, stdout=so, stderr=se,
                                     start_new_session=True, cwd=os.getcwd())
        last_size, last_change, my, why = 0, time.time(), 0, ""      # my: 0 running, 1 finished well, 2 failed
        while True:
            if my == 0:
                rc = child.poll()
                sz = os.path.getsize(hb)
                if sz != last_size:
                    last_size, last_change = sz, time.time()
                if rc is not None:
                    my, why = (1, "") if rc in (0, 4) else (2, f"exit code {rc}")
                    if rc == 4:
                        why = "decomposition-invariance check failed"
                elif time.time() - last_change > args.stall_timeout:
                    my, why = 2, f"no progress for {args.stall_timeout:.0f} s after phase '{open(hb).read().strip().splitlines()[-1:] or ['start']}'"
                elif time.time() - t0 > args.attempt_timeout:
                    my, why = 2, f"attempt longer than {args.attempt_timeout:.0f} s"
            t = torch.tensor([int(my == 0), int(my == 1), int(my == 2), int(my == 1 and child.returncode == 4)], dtype=torch.int64)
            dist.all_reduce(t)
            if int(t[2]) > 0 or int(t[1]) == world:
                break
            time.sleep(0.5)
        failed, inv_failed = int(t[2]) > 0, int(t[3]) > 0
        if child.poll() is None:      # somebody's child failed or hung: this attempt is over for everybody
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            child.wait()
        reasons = [None] * world
        dist.all_gather_object(reasons, why)
        att = {"rung": ri, "name": name, "env": env_extra, "seconds": round(time.time() - t0, 1),
               "outcome": "failed" if failed else ("ran, decomposition-invariance check FAILED" if inv_failed else "ok"),
               "reasons": sorted(set(r for r in reasons if r))}
        attempts.append(att)
        got = None
        if rank == 0 and not failed:
            got = next((ln for ln in open(fo).read().splitlines() if ln.startswith('{"metric"')), None)
        if failed or (rank == 0 and got is None):
            tail = open(fe).read()[-3000:]
            print(f"[bench supervisor rank {rank}] rung {ri} ({name}) failed: {why or 'another rank failed'}\n{tail}", file=sys.stderr, flush=True)
        ok = [bool(got) and not inv_failed]
        dist.broadcast_object_list(ok, src=0)
        if rank == 0 and got and fallback_line is None:
            fallback_line = (ri, got)
        if ok[0]:
            line = (ri, got)
            break
    if rank == 0:
        use = line or fallback_line
        if use:
            d = json.loads(use[1])
            d["ladder"] = {"rung": use[0], "name": rungs[use[0]][0], "env": rungs[use[0]][1], "attempts": attempts,
                           "outcome": "ok" if line else "NO rung passed the decomposition-invariance check: the number is of the first rung that ran to its end",
                           "stall_timeout_s": args.stall_timeout, "attempt_timeout_s": args.attempt_timeout}
            print(json.dumps(d), flush=True)
        else:
            print(f"[bench supervisor] no rung of the ladder ran to its end: {json.dumps(attempts)}", file=sys.stderr, flush=True)
    good = [bool(line or fallback_line)]
    dist.broadcast_object_list(good, src=0)
    dist.destroy_process_group()
    raise SystemExit(0 if good[0] else 5)


def live_traffic(args, dom, timeout_s=240):
    """HBM bytes per launch of the dominant kernel, measured in THIS run: two child runs of this script (the same workload, 9 substeps,
    no CPU / drop-in legs) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- one counter per pass and
    kernel tracing only, as MI355X_MICROARCH.md prescribes (counters serialise the kernels, so they cannot sit in the timed region);
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 with the guide's gfx950 correction of the fetch tally (profiles/tools/summarise.py
    applies the same formula to the kept profiles).  None when rocprofv3 is absent, this process is itself being profiled, or a pass
    fails: the caller then falls back to the committed collection and says so."""
    import csv
    import glob
    import shutil
    pats = {"closure": "closure_lds_kernel", "div_rhs": "div_rhs_kernel", "thomas": "thomas_", "project_integrate": "integrate_kernel",
            "scalar_kappa_faces": "scalar_kappa_faces_kernel", "scalar_lds_cd2": "scalar_lds_kernel", "scalar": "scalar_kernel"}
    pat = "mom_lds_kernel" if dom.startswith("mom_") else pats.get(dom)
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if pat is None or prof is None or any(k.startswith("ROCPROF") for k in os.environ):
        return None
    flags = ["--no-cpu", "--no-dropin", "--no-pmc", "--steps", "6", "--warmup", "3", "--sgs", args.sgs, "--nsv", str(args.nsv)]
    if args.size:
        flags += ["--size", args.size]
    if args.ibm:
        flags.append("--ibm")
    if args.no_floor:
        flags.append("--no-floor")
    means, launches = {}, 0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [prof, "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__)] + flags
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except Exception:      # noqa: BLE001
                return None
            if r.returncode != 0:
                return None
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and pat in row.get("Kernel_Name", ""):
                            tot += float(row["Counter_Value"])
                            n += 1
            if n == 0:
                return None
            means[counter], launches = tot / n, n
    return {"bytes": int((2.0 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024), "launches": launches,
            "FETCH_SIZE_KiB": round(means["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(means["WRITE_SIZE"], 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--size", type=str, default="", help="override grid, e.g. 256x256x256")
    ap.add_argument("--weak", action="store_true", help="N>1: one 256^3 slab per GPU (weak scaling) instead of 1024x512x512 split N ways")
    ap.add_argument("--no-single", action="store_true", help="N>1: skip rank 0's one-GPU run of the same grid")
    ap.add_argument("--with-single", action="store_true", help="N=1: run that leg anyway (exercises the code path on a one-GPU box)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-dropin", action="store_true", help="skip the Fortran drop-in leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes behind roofline.traffic (N = 1)")
    ap.add_argument("--nsv", type=int, default=0, help="passive scalars (kappa scheme), BASELINE configs[2]")
    ap.add_argument("--sgs", type=str, default="vreman", choices=["vreman", "smag"])
    ap.add_argument("--oversubscribe", action="store_true",
                    help="testing only: let WORLD_SIZE exceed the node's GPUs (ranks share devices, rendezvous over gloo); "
                         "RCCL refuses two ranks on one device, so the run stops cleanly at udc_comm_init")
    ap.add_argument("--ibm", action="store_true",
                    help="a staggered cube array with the immersed boundary and neutral facet wall functions (BASELINE configs[4]'s layout)")
    ap.add_argument("--no-floor", action="store_true",
                    help="free floor instead of the neutral log-law wall function (lbottom, BCbotm = 3) of SURVEY 8d")
    ap.add_argument("--no-invariance", action="store_true", help="N>1: skip the field comparison with the one-GPU run")
    ap.add_argument("--invariance-substeps", type=int, default=6, help="substeps from the cold start after which the fields are compared")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)      # the supervised child of an N>1 run
    ap.add_argument("--no-ladder", action="store_true", help="N>1: one attempt with the defaults, no fallback rungs")
    ap.add_argument("--stall-timeout", type=float, default=240., help="N>1: seconds without a heartbeat before an attempt is killed")
    ap.add_argument("--attempt-timeout", type=float, default=1500., help="N>1: seconds an attempt may take in all")
    args = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.worker and int(os.environ.get("WORLD_SIZE", "1")) == args.gpus:
        supervise(args, sys.argv[1:])
    heartbeat("start")

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libudcore has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev and not args.oversubscribe:
        raise SystemExit(f"--gpus {world}: this node has {ndev} GPU(s); one rank per GPU (RCCL refuses two ranks on one device)")
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    if world > 1:
        if args.oversubscribe:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    from udcore import read_deck, cold_start
    from udcore.grid import Grid
    import udcore

    if args.size:
        nx, ny, nz = (int(x) for x in args.size.lower().split("x"))
    elif world == 1:
        nx, ny, nz = 256, 256, 256              # BASELINE.json configs[1]
    elif args.weak:
        nx, ny, nz = 256, 256 * world, 256      # weak scaling: one 256^3 slab of the channel per GPU
    else:
        nx, ny, nz = 1024, 512, 512             # BASELINE.json configs[3]: strong scaling, the same grid at every N
    scaling = "weak" if (args.weak and not args.size) else "strong"
    dt = 0.25
    with tempfile.TemporaryDirectory() as tmp:
        deck = read_deck(write_deck(tmp, 901, nx, ny, nz, 0, dt=dt, nprocy=world, nsv=args.nsv, sgs=args.sgs, floor=not args.no_floor))
    core = udcore.from_deck(deck, device=dev, rank=rank, nranks=world)
    if world > 1:
        iddev = "cpu" if args.oversubscribe else "cuda"
        idt = torch.zeros(128, dtype=torch.uint8, device=iddev)
        if rank == 0:
            import ctypes
            buf = (ctypes.c_ubyte * 128)()
            core.lib.udc_comm_unique_id(buf)
            idt = torch.tensor(list(buf), dtype=torch.uint8, device=iddev)
        dist.broadcast(idt, 0)
        try:
            core.comm_init(bytes(idt.cpu().tolist()))
        except RuntimeError as e:
            # e.g. two ranks on one device: RCCL refuses, nothing has been exchanged yet -- leave without a collective
            print(f"[bench rank {rank}] udc_comm_init refused: {e}", file=sys.stderr, flush=True)
            core.close()
            dist.destroy_process_group()
            raise SystemExit(3)
    g = core.g
    nyl = ny // world
    ibm_counts = cube_array_ibm(core, nx, ny, nz) if args.ibm else None
    heartbeat("communicator")
    st = cold_start(g, deck, j0=rank * nyl, nyl=nyl, nsv=args.nsv)
    core.load_state(st)
    core.halos()
    core.boundary()
    want_single = (world > 1 and not args.no_single) or args.with_single
    want_inv = want_single and not args.no_invariance and args.invariance_substeps > 0
    nsub_inv = 3 * ((args.invariance_substeps + 2) // 3)      # whole RK3 steps: the timed loop starts at stage 1 again
    init_rows = {k: np.ascontiguousarray(st[k][1:-1, 1:-1, 1:-1]) for k in ("u0", "v0", "w0")} if want_inv else None
    del st
    heartbeat("cold start loaded")
    after_rows = None
    if want_inv:
        # decomposition invariance, first half: the slabs' fields after `nsub_inv` substeps from the cold start (the one-GPU run they
        # are compared with follows after the timed region, on rank 0)
        rk_ = 1
        for _ in range(nsub_inv):
            core.substep(rk_, dt, True)
            rk_ = rk_ % 3 + 1
        core.sync()
        after_rows = {k: np.ascontiguousarray(core.download(k)[1:-1, 1:-1, 1:-1]) for k in ("u0", "v0", "w0", "pres0")}
        heartbeat("invariance substeps")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        core.sync()

    # Warm-up doubles as the per-kernel survey: after a few unmarked substeps every launch is bracketed by HIP events
    # (one chained event per launch boundary; each costs ~6 us on the GPU timeline, ~4 % of a 256^3 substep) -> the kernel
    # table and the dominant kernel.  The timed region then carries events around that kernel only (two per substep), so
    # `value` is not taxed by the survey and roofline.achieved is still measured live inside the timed region.
    # The survey covers at least six substeps (two of every RK stage) whatever --warmup says: with a short warm-up the missing ones
    # are run on top of it, untimed like the warm-up itself.
    rk = 1
    skip = min(3, args.warmup)
    for _ in range(skip):
        core.substep(rk, dt, True)
        rk = rk % 3 + 1
    table, n_tab, tab_stage1 = {}, (max(args.warmup - skip, 6) if args.warmup > 0 else 0), 0
    if n_tab > 0:
        core.sync()
        core.profile(True)
        core.profile_reset()
        for _ in range(n_tab):
            core.substep(rk, dt, True)
            tab_stage1 += rk == 1
            rk = rk % 3 + 1
        core.sync()
        table = fold_edges(core.profile_get())
        core.profile(False)
    nscal = args.nsv                      # transported scalars the integrate kernel also advances (the bench deck has no thl, qt)
    cand = [k for k in table if algo_bytes(k, nscal)]
    dom = max(cand, key=lambda k: table[k][0]) if cand else None
    heartbeat("warm-up")
    barrier()
    # the dominant kernel's launches carry their two events in every 4th substep of the timed region (the RK stages take turns): the
    # pair costs the stream ~11 us, 1.2 % of a 256^3 substep if every launch carried it -- `value` is taxed by a quarter of that
    every = 4 if (args.steps >= 24 and dom) else 1
    core.profile(True, focus=dom, every=every)
    core.profile_reset()
    stage1 = timed_substeps = 0
    t0 = time.perf_counter()
    for it in range(args.steps):
        core.substep(rk, dt, True)
        if it % every == 0:
            stage1 += rk == 1
            timed_substeps += 1
        rk = rk % 3 + 1
    barrier()
    t1 = time.perf_counter()
    prof = fold_edges(core.profile_get())
    core.profile(False)
    elapsed = t1 - t0

    def allmax(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    heartbeat("timed region")
    elapsed = allmax(elapsed)
    divmax, _ = core.divergence()
    # the reference's `poisson` routine alone (src/modpois.f90:419: fillps + bcpup, FFTs / transposes, solmpj, tderive + bcp)
    core.rk3step, core.dt = 1, dt
    for _ in range(3):
        core.poisson()
    poisson_ms = allmax(time_loop(core, core.poisson, 20, barrier))
    heartbeat("poisson alone")
    executed_plan = core.last_plan()
    global PRESSURE_TOTAL, SCALARS_INLINE
    PRESSURE_TOTAL = bool(executed_plan.get("pressure_total_form"))
    SCALARS_INLINE = int(executed_plan.get("scalars_updated_in_their_sweep", 0))
    # What the exchanges did (slab layout only).  (i) the communicator's own account of itself; (ii) six more substeps with every
    # exchange counted and timed by a pair of events on the stream it runs on: bytes per peer, and the rate one link saw during an
    # all-to-all (every peer's block travels at once, so a block's bytes / the operation's time is the per-link rate); (iii) the same
    # substeps with the exchanges switched off (udc_comm_dry_run: kernels only, results wrong -- this state is not used again):
    # substep - that = the exchange time the overlap did NOT hide.
    exchange = None
    if executed_plan["slab_layout"] and os.environ.get("UDC_BENCH_NO_EXCHANGE_ACCOUNT", "0") in ("", "0"):
        try:
            nx_sub = 6
            core.comm_stats(1)
            for _ in range(nx_sub):
                core.substep(rk, dt, True)
                rk = rk % 3 + 1
            barrier()
            cs = core.comm_stats(2)
            a2a_ms_op = cs["alltoall_ms"] / cs["alltoall_ops"] if cs["alltoall_ops"] else None
            core.comm_dry_run(True)
            for _ in range(3):
                core.substep(rk, dt, True)
                rk = rk % 3 + 1
            n_dry = max(6, min(args.steps, 30))
            barrier()
            td0 = time.perf_counter()
            for _ in range(n_dry):
                core.substep(rk, dt, True)
                rk = rk % 3 + 1
            barrier()
            dry_ms = allmax((time.perf_counter() - td0) / n_dry * 1e3)
            core.comm_dry_run(False)
            exchange = {
                "alltoall_per_substep": cs["alltoall_ops"] / nx_sub, "alltoall_bytes_per_peer": int(cs["alltoall_bytes_per_peer"]),
                "alltoall_bytes_sent_per_substep": int(cs["alltoall_bytes_sent"] / nx_sub),
                "alltoall_ms_per_operation": round(a2a_ms_op, 5) if a2a_ms_op else None,
                "alltoall_GBs_per_link": round(cs["alltoall_bytes_per_peer"] / (a2a_ms_op * 1e-3) / 1e9, 6) if a2a_ms_op else None,
                "ghost_row_exchanges_per_substep": cs["ghost_row_exchanges"] / nx_sub,
                "ghost_row_bytes_to_prev_per_substep": int(cs["ghost_row_bytes_to_prev"] / nx_sub),
                "ghost_row_bytes_to_next_per_substep": int(cs["ghost_row_bytes_to_next"] / nx_sub),
                "ghost_row_ms_per_substep": round(cs["ghost_row_ms"] / nx_sub, 5),
                "allreduce_per_substep": cs["allreduce_ops"] / nx_sub,
                "substep_ms_exchanges_off": round(dry_ms, 5),
                "exposed_exchange_ms": round(elapsed / args.steps * 1e3 - dry_ms, 5),
                "note": "rank 0's counters over 6 substeps after the timed region; times from event pairs on the exchange's own stream; "
                        "exposed = timed substep - the same substep with every exchange a no-op (max over ranks)"}
        except Exception as e:      # noqa: BLE001 (a side measurement)
            exchange = {"error": repr(e)[:300]}
        heartbeat("exchange account")
    # the same solve as the fused substep runs it (pup mode: the divergence of the stored predicted velocity, on the slab
    # path inside the x transform; projection fused with the RK3 update): substep time minus its non-Poisson kernels
    survey = table if table else prof          # (no warm-up: the timed region carried every marker)
    n_survey = n_tab if table else args.steps
    nonpois = sum(ms for name, (ms, cnt) in survey.items() if name.startswith(("closure", "mom_", "bottom", "scalar"))) / max(n_survey, 1)
    poisson_in_substep_ms = elapsed / args.steps * 1e3 - nonpois

    cells = nx * ny * nz
    cells_local = nx * nyl * nz
    value = cells * args.steps / elapsed
    kernels = {}
    ms_per = elapsed / args.steps * 1e3
    # what a marker costs: every surveyed launch sits between two chained events (~6 us each on the GPU timeline), so the surveyed
    # durations add up to more than an unmarked substep takes; the difference, per launch, is taken back out of the surveyed entries
    # (`avg_ms` stays what the events said, `avg_ms_net` and `share` are net of it; the dominant kernel's figure comes from the
    # timed region, where it alone is marked)
    n_launch = sum(cnt for _, (ms, cnt) in table.items()) / max(n_tab, 1) if table else 0
    s_survey = sum(ms for _, (ms, cnt) in table.items()) / max(n_tab, 1) if table else 0.
    marker_ms = max(0., (s_survey - ms_per) / n_launch) if n_launch else 0.
    for name, (ms, cnt) in survey.items():
        live = table and name in prof          # the dominant kernel: measured inside the timed region
        if live:
            ms, cnt = prof[name]
        s1 = (stage1 / max(timed_substeps, 1)) if (live or not table) else (tab_stage1 / max(n_tab, 1))
        ab = algo_bytes(name, nscal, s1)
        if name == "fftx_pack_fwd" and "div_rhs" not in survey:
            ab += 16                           # the divergence is folded into this stage (ALGO_BYTES)
        avg_ms = ms / max(cnt, 1)
        per_substep = cnt / max(timed_substeps if (live or not table) else n_tab, 1)
        net = avg_ms if (live or not table) else max(avg_ms - marker_ms, 0.)
        ent = {"avg_ms": round(avg_ms, 5), "avg_ms_net": round(net, 5), "launches": cnt, "launches_per_substep": round(per_substep, 3),
               "share": round(net * per_substep / ms_per, 4),
               "measured": (f"timed region, every {every}th substep" if every > 1 else "timed region") if (live or not table) else f"survey over {n_tab} untimed substeps, every launch marked"}
        if ab and net > 0.:      # (a launch shorter than the marker's cost nets to zero on tiny test grids: no rate for it)
            gbs = ab * cells_local / (net * 1e-3) / 1e9
            ent.update({"algo_bytes_per_cell": round(ab, 2), "achieved_GBs": round(gbs, 1),
                        "frac": round(gbs / HBM_PEAK_GBS, 4)})
        kernels[name] = ent
    if dom is None or dom not in kernels:
        dom = max((k for k in kernels if "frac" in kernels[k]), key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
    # HBM traffic of the dominant kernel from the committed PMC collection (profiles/pmc_traffic.json, written by
    # profiles/tools/collect.sh from separate rocprofv3 --pmc passes: counters cannot be read inside a timed run);
    # entries are keyed by workload (grid per GPU, SGS, scalars) and by kernel, so a stale or foreign entry never matches
    traffic, traffic_src = None, None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            with open(tfile) as f:
                tj = json.load(f)
            key = f"{nx}x{nyl}x{nz}/{args.sgs}/nsv{args.nsv}"
            ent = tj.get("workloads", {}).get(key, {}).get(dom)
            if ent:
                traffic, traffic_src = ent["hbm_bytes_per_launch"], ent.get("source")
        except Exception:
            traffic = None
    # ... unless this run can measure it itself (one GPU, rocprofv3 on the box): two short child passes after the timed region
    if world == 1 and not args.no_pmc:
        lt = live_traffic(args, dom)
        if lt:
            traffic = lt["bytes"]
            traffic_src = (f"this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of the same workload, mean of "
                           f"{lt['launches']} launches, (2 x {lt['FETCH_SIZE_KiB']} + {lt['WRITE_SIZE_KiB']}) KiB")
        elif traffic is not None:
            traffic_src = f"{traffic_src} (committed collection: no live counter pass in this run)"
    roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["frac"], "traffic": traffic, "traffic_source": traffic_src,
                "algo_bytes_per_launch": int(kernels[dom]["algo_bytes_per_cell"] * cells_local),
                "avg_launch_ms": kernels[dom]["avg_ms"],
                # the rate MI355X_MICROARCH.md calls achievable for streaming kernels, next to the nominal peak
                "achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": round(kernels[dom]["achieved_GBs"] / HBM_ACHIEVABLE_GBS, 4)}
    # measured copy ceiling of this GPU (BASELINE.md asks for the fraction against it next to the nominal 8 TB/s):
    # a 1 GiB device-to-device copy, read + write bytes over the best of 10 repetitions, outside the timed region
    try:
        src = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
        dst = torch.empty_like(src)
        best = float("inf")
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dst.copy_(src); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        ceiling = 2 * src.numel() * 8 / (best * 1e-3) / 1e9
        roofline["copy_ceiling"] = round(ceiling, 1)
        roofline["frac_of_copy_ceiling"] = round(kernels[dom]["achieved_GBs"] / ceiling, 4)
        roofline["copy_ceiling_note"] = "1 GiB device-to-device copy on this GPU, read + write bytes, best of 12"
        del src, dst
    except Exception:      # noqa: BLE001 (the ceiling is a side measurement: never let it take the bench line down)
        pass
    # bytes the substep as built must move: every kernel's algorithmic bytes x the whole-grid passes it makes per substep -- its
    # launches per substep on the single-slab path (one per transported scalar for the scalar sweeps); on the slab path the sweeps
    # are cut into launches over parts of the slab (edge rows / interior, level ranges) that add up to one pass
    slab_path = world > 1 or os.environ.get("UDC_FORCE_SLAB", "0") not in ("", "0")
    ab_sum = sum(k["algo_bytes_per_cell"] * ((max(args.nsv, 1) if name.startswith("scalar") else 1) if slab_path else k["launches_per_substep"])
                 for name, k in kernels.items() if "algo_bytes_per_cell" in k)
    as_built = {"bytes_per_cell_update": round(ab_sum, 1),
                "frac_of_hbm_peak": round(ab_sum * cells_local * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4)}
    as_built["frac_of_achievable"] = round(ab_sum * cells_local * args.steps / elapsed / 1e9 / HBM_ACHIEVABLE_GBS, 4)
    if "copy_ceiling" in roofline:
        as_built["frac_of_copy_ceiling"] = round(ab_sum * cells_local * args.steps / elapsed / 1e9 / roofline["copy_ceiling"], 4)
    out = {
        "metric": "cell-updates/sec (advect+diffuse+Poisson step)", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_name(nx, ny, nz, args.sgs, args.nsv, not args.no_floor, args.ibm),
                   **({"immersed_boundary": ibm_counts} if args.ibm else {}),
                   "floor": "free (no wall function)" if args.no_floor else
                            "neutral log-law wall function (lbottom, BCbotm=3, z0=0.05)",
                   "grid": [nx, ny, nz], "decomposition": f"y-slabs x{world}", "dt": dt,
                   # the order the substeps ran in, as the library's planner decided it (udc_last_plan), not as the environment suggests
                   "executed_plan": executed_plan,
                   "step": "one RK3 substep = one cell-update per cell"},
        # SURVEY section 8(d)'s byte MODEL of a substep (392 B per cell, each routine of the reference on its own) over the measured time
        # and the 8 TB/s peak: a ratio for continuity with rounds 1-4, NOT a bandwidth (the substep as built moves a third less)
        "model_392B_ratio": round(392.0 * cells_local * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4),
        # the same on the bytes the kernels as built must move (sum of the per-kernel algorithmic bytes, launches per
        # substep as surveyed), against the 8 TB/s peak and against this box's measured copy rate
        "whole_substep_as_built": as_built,
        "divmax_after_run": divmax,
        "poisson_only_ms": round(poisson_ms, 5),
        "poisson_in_substep_ms": round(poisson_in_substep_ms, 5),
        "poisson_note": "poisson_only = udc_poisson on its own, the routine-by-routine entry point a driver in UDC_RESIDENCY 0 / 1 calls (divergence, "
                        "solve and projection as separate kernels, the reference's p / pres0 form); poisson_in_substep = the fused substep minus its "
                        "sweeps (divergence inside the x transform, projection fused with the integration, pressure-total form)",
        "roofline": roofline,
        "kernels": kernels,
        "kernels_note": {"marker_cost_ms_per_launch": round(marker_ms, 5), "surveyed_substeps": n_tab,
                         "sum_of_shares": round(sum(k["share"] for k in kernels.values()), 4)},
    }
    if executed_plan["slab_layout"]:
        out["rccl"] = core.comm_info()
        out["exchange"] = exchange
    inv_ok = True
    if want_single:
        # rank 0's own one-GPU run of the SAME grid (single-slab code path), so that every N>1 line carries its strong-scaling
        # reference AND the field comparison behind it; the other ranks hand over their rows and wait at the barrier below
        single, inv = None, None
        init_g = after_g = None
        if want_inv:
            def gather_rows(a):      # [nz, nyl, nx] of every rank -> [nz, ny, nx] on rank 0 (device tensors under RCCL, host under gloo)
                t = torch.from_numpy(a)
                if world == 1:
                    return t
                t = t.cuda() if dist.get_backend() == "nccl" else t
                parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
                dist.gather(t, parts, dst=0)
                return torch.cat(parts, dim=1) if rank == 0 else None
            init_g = {k: gather_rows(v) for k, v in init_rows.items()}
            after_g = {k: gather_rows(v) for k, v in after_rows.items()}
            del init_rows, after_rows
            heartbeat("rows gathered")
        if rank == 0:
            try:
                single, inv = single_gpu_leg(nx, ny, nz, dt, args, dev, elapsed / args.steps * 1e3,
                                             poisson_ms, poisson_in_substep_ms, init_g if want_inv else None, after_g, nsub_inv)
            except Exception as e:      # noqa: BLE001 (a side measurement: never let it take the bench line down)
                single = {"error": repr(e)[:300]}
        del init_g, after_g
        if world > 1:
            dist.barrier()
        heartbeat("one-GPU leg")
        out["single_gpu_same_workload"] = single
        if world > 1 and isinstance(single, dict) and "speedup_substep" in single:
            # strong scaling at a glance (the north star states its >= 6x on the Poisson solve): one GPU's time over this run's, same grid
            out["strong_scaling_vs_one_gpu"] = {"substep": single["speedup_substep"], "poisson_only": single["speedup_poisson"],
                                                "poisson_in_substep": single["speedup_poisson_in_substep"], "n_gpus": world}
        if want_inv:
            out["decomposition_invariance"] = inv if inv else {"ok": False, "error": "the one-GPU run did not finish", "tolerance": INVARIANCE_TOL}
            inv_ok = bool(inv and inv["ok"])
    if rank == 0:
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(nx, ny, nz)
            out["cpu_baseline"] = cb if cb else {"value": None, "unit": "cell-updates/s", "cores": 0,
                                                 "kind": "reference", "sample": "oracle/_ref/udales_ref unavailable"}
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_dropin:
            core.close()
            core = None
            out["dropin"] = dropin_leg(nx, ny, nz, args.nsv, args.sgs, not args.no_floor, value)
        print(json.dumps(out))
    if core is not None:
        core.close()
    if world > 1:
        ok = [inv_ok]
        dist.broadcast_object_list(ok, src=0)
        inv_ok = ok[0]
        dist.destroy_process_group()
    heartbeat("done")
    if not inv_ok:
        raise SystemExit(4)      # the line is printed, flagged; a supervised run goes on to the next rung


if __name__ == "__main__":
    main()
