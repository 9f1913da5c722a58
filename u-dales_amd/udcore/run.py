"""Command-line runner: the time loop of src/program.f90:126-222 over the device core.

    python -m udcore.run namoptions.NNN [--steps N] [--restart-from NTRUN] [--device D] [--quiet]

reads the deck (namoptions, prof.inp, lscale.inp) from the file's directory, cold-starts (or warm-starts from the
reference's initd/inits restart files with --restart-from NTRUN; or the deck's own &RUN lwarmstart / startfile), advances until `runtime` (or N full steps)
with the reference's own time-step control (tstep_update, src/modtstep.f90:113-150: fixed dtmax, or adaptive with the
Courant / diffusion numbers), and writes restart files in the reference's layout every `trestart` seconds of model
time and at the end (src/modsave.f90:77-121).  Immersed boundaries (libm) and non-periodic lateral
boundaries are not available on the device path: such decks are refused with the reference's error convention.

Multi-GPU: launch one process per GPU with torch.distributed.run; the y-slab decomposition follows WORLD_SIZE.
"""
from __future__ import annotations

import argparse
import os
import sys
import time


def _refuse(msg):
    sys.stderr.write(f" ERROR: {msg}\n")
    raise SystemExit(1)


def check_supported(deck):
    from .namoptions import UNSUPPORTED
    g = deck.get
    for grp, name, off in UNSUPPORTED:                       # features without a device implementation
        if (grp, name) == ("DRIVER", "idriver") and deck.is_set(grp, name) and int(g(grp, name)) == 2 and int(g("BC", "BCxm")) == 3 \
                and not g("DRIVER", "lchunkread"):
            continue      # (reading a precursor's planes: udcore.driver; writing them, idriver = 1, stays with the reference)
        if deck.is_set(grp, name) and deck.nml[grp][[k for k in deck.nml[grp] if k.lower() == name.lower()][0]] != off:
            _refuse(f"&{grp} {name} is not available on the device path")
    if int(g("DYNAMICS", "iadv_mom")) != 2:
        _refuse("Unknown advection scheme: only iadv_mom = 2 (cd2) is on the device path")      # src/modadvection.f90:52
    if deck.is_set("BC", "BCxs") and int(g("BC", "BCxs")) not in ((1, 2, 3) if int(g("BC", "BCxm")) == 3 else (1, 2)):
        _refuse("&BC BCxs: only 1 (periodic), 2 (inflow profile, convective outflow) and, with BCxm = 3, 3 (driver planes) are on the device path")
    for grp, names in (("BC", ("BCxT", "BCxq", "BCyT", "BCyq", "BCys")),):
        for n in names:
            if n in ("BCxT", "BCxq") and int(g("BC", "BCxm")) in (2, 3) and int(g("BC", n)) in ((1, 2, 3) if int(g("BC", "BCxm")) == 3 else (1, 2)):
                continue      # (inflow / outflow for the flow and the temperature: udc_create_open_x, udc_set_open_x_thl)
            if deck.is_set(grp, n) and int(deck.nml[grp][[k for k in deck.nml[grp] if k.lower() == n.lower()][0]]) != 1:
                _refuse(f"only periodic lateral boundaries are on the device path (&BC {n})")
    # (iwallmom = 2 without the temperature equation never reaches initibm: checkinitvalues has made it 3 by then,
    #  src/modstartup.f90:811-816 -- Deck.apply_checkinitvalues)
    if int(g("BC", "BCxm")) not in (1, 2, 3) or int(g("BC", "BCym")) != 1:
        _refuse("lateral boundaries on the device path: BCxm = 1 (periodic), 2 (inflow profile, convective outflow) or 3 (a precursor's planes), BCym = 1")
    if int(g("DYNAMICS", "ipoiss")) != 0 or int(g("BC", "BCzp")) not in (1, 2):
        _refuse("only ipoiss = 0 (FFT in x, y) with BCzp = 1 or 2 is on the device path")
    # &RUN nprocx / nprocy describe the CPU run's pencil layout; the device path splits y over however many GPUs it is
    # launched on and the fields do not depend on it.  (One thing in the reference does: the immersed boundary's point masks
    # wrap periodically only in a direction that is split over ranks -- udcore.ibm hands the deck's values to
    # udc_set_ibm_mask_wrap so that the run reproduces the reference run of this deck.)


def courant_default(deck):
    """src/modglobal.f90:563-577."""
    c = float(deck.get("RUN", "courant"))
    if c >= 0:
        return c
    iadv_mom = int(deck.get("DYNAMICS", "iadv_mom"))
    c = 1.5 if iadv_mom == 2 else 1.4
    # scalars always use the kappa scheme (:557-559); thl, qt, tke inherit the momentum scheme when < 0 (:549-551)
    adv = [iadv_mom if int(deck.get("DYNAMICS", n)) < 0 else int(deck.get("DYNAMICS", n)) for n in ("iadv_thl", "iadv_qt", "iadv_tke")]
    sv = [7] * int(deck.get("SCALARS", "nsv"))
    if 7 in sv + adv or 1 in sv + adv:
        c = min(c, 1.1)
    elif 2 in sv + adv:
        c = min(c, 1.5)
    return c


class FieldDump:
    """Instantaneous 3-D fields every tfielddump seconds (src/modfielddump.f90: &OUTPUT lfielddump, tfielddump, fieldvars).
    The reference writes NetCDF-4 `fielddump.xxx.xxx.expnr.nc` per rank; here `fielddump.<rank>.<expnr>.npz`, one array per
    variable and record (float32 like the reference's output), variables named as in fieldvars: u0, v0, w0, th (thl0),
    ql / qt (qt0), p0 (pres0), s1.. (scalars).  Interior cells kb..ke."""
    NAMES = {"u0": "u0", "v0": "v0", "w0": "w0", "th": "thl0", "qt": "qt0", "p0": "pres0"}

    def __init__(self, core, tfielddump, fieldvars, wdir, expnr, rank=0):
        from . import lib as L
        self.core, self.dt, self.wdir, self.expnr, self.rank = core, float(tfielddump), wdir, expnr, rank
        self.vars = [v.strip() for v in str(fieldvars).split(",") if v.strip()]
        for v in self.vars:
            if v not in self.NAMES and not (v[0] == "s" and v[1:].isdigit() and 1 <= int(v[1:]) <= core.nsv):
                _refuse(f"&OUTPUT fieldvars: '{v}' is not available on the device path (u0 v0 w0 th qt p0 s1..)")
        self.L, self.tnext, self.times, self.rec = L, self.dt, [], {v: [] for v in self.vars}

    def step(self, timee):
        if timee < self.tnext:                                  # src/modfielddump.f90: tnextfielddump
            return False
        self.tnext += self.dt
        import numpy as np
        for v in self.vars:
            if v in self.NAMES:
                a = self.core.download(self.NAMES[v])[1:-1, 1:-1, 1:-1]
            else:
                a = self.core.download(self.L.scalar_field(self.L.SV0, int(v[1:]) - 1), halo=2)[2:-2, 2:-2, 2:-2]
            self.rec[v].append(a.astype(np.float32))
        self.times.append(timee)
        np.savez(os.path.join(self.wdir, f"fielddump.{self.rank:03d}.{self.expnr:03d}.npz"), time=np.array(self.times),
                 **{v: np.array(r) for v, r in self.rec.items()})
        return True


def main(argv=None, at_end=None):
    """at_end(core, tdump): called before the core is closed (tests look at the state and the statistics there)."""
    ap = argparse.ArgumentParser(prog="python -m udcore.run", description=__doc__.split("\n\n")[0])
    ap.add_argument("namoptions")
    ap.add_argument("--steps", type=int, default=0, help="stop after this many full time steps (default: run to `runtime`)")
    ap.add_argument("--restart-from", type=int, default=-1, metavar="NTRUN",
                    help="warm start from initd<NTRUN>_000_<rank>.<expnr> in the deck's directory")
    ap.add_argument("--ibm-mask-wrap", choices=("deck", "none", "both"), default="deck",
                    help="immersed boundary: the reference's point masks wrap periodically only in a direction its run splits over ranks "
                         "(&RUN nprocx, nprocy), which decides what an obstacle cell on the edge of the domain averages over; "
                         "'deck' reproduces the reference run with the deck's layout, 'none' its single-rank run, 'both' a fully split one")
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)

    from . import cold_start, from_deck, read_deck
    from . import restart as R
    from .forcings import LevelForcings

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    deck = read_deck(args.namoptions)
    if args.ibm_mask_wrap != "deck":      # (in memory only: the device's own decomposition does not come from these)
        for key in ("nprocx", "nprocy"):
            deck.set("RUN", key, 1 if args.ibm_mask_wrap == "none" else 2)
    check_supported(deck)
    wdir = os.path.dirname(os.path.abspath(args.namoptions))
    core = from_deck(deck, device=device, rank=rank, nranks=world)
    if world > 1:
        import ctypes
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf = (ctypes.c_ubyte * 128)()
            core.lib.udc_comm_unique_id(buf)
            idt = torch.tensor(list(buf), dtype=torch.uint8, device="cuda")
        dist.broadcast(idt, 0)
        core.comm_init(bytes(idt.cpu().tolist()))
    iexp = int(deck.get("RUN", "iexpnr"))
    dtmax = float(deck.get("RUN", "dtmax"))
    runtime = float(deck.get("RUN", "runtime"))
    trestart = float(deck.get("RUN", "trestart"))
    ladaptive = bool(deck.get("RUN", "ladaptive"))
    courant, diffnr = courant_default(deck), float(deck.get("RUN", "diffnr"))
    nyl = core.g.ny // world
    warm = args.restart_from
    if warm < 0 and bool(deck.get("RUN", "lwarmstart")):      # &RUN lwarmstart / startfile = 'initdNNNNNNNN_xxx_xxx.EEE' (src/modstartup.f90:784, 2194)
        import re
        m = re.search(r"initd(\d{8})_", str(deck.get("RUN", "startfile")))
        if not m:
            _refuse("&RUN lwarmstart: startfile must be named initd<ntrun>_xxx_xxx.<expnr>")
        warm = int(m.group(1))
    if warm >= 0:
        # the restart set may come from any CPU pencil layout (nprocx x nprocy files); scalars are read when the deck says so
        # (lreadscal, :2212) or when the warm start was asked for on the command line
        rs = bool(deck.get("SCALARS", "lreadscal")) or args.restart_from >= 0
        timee, dt = R.load_restart_global(core, wdir, iexp, warm, rank=rank, nranks=world, read_scalars=rs)
        if core.nsv and not rs:      # scalars start from their profile (:2220-2229)
            stc = cold_start(core.g, deck, j0=rank * nyl, nyl=nyl, nsv=core.nsv)
            from . import lib as L
            for n in range(core.nsv):
                core.upload(L.scalar_field(L.SV0, n), stc[f"sv0_{n}"]); core.upload(L.scalar_field(L.SVM, n), stc[f"svm_{n}"])
            core.halos(); core.boundary()
        ntrun = warm
    else:
        core.load_state(cold_start(core.g, deck, j0=rank * nyl, nyl=nyl, nsv=core.nsv, pre_boundary=True))
        core.halos()
        timee, ntrun = 0., 0
        dt = dtmax if not ladaptive else dtmax / 100.          # src/modstartup.f90:1099, 2038
    core.dt, core.timee, core.rk3step = dt, timee, 0
    inlet = None
    if getattr(core, "driver_inflow", False):      # &BC BCxm = 3: the precursor's planes (src/moddriver.f90 readdriverfile; modstartup.f90:1462-1470)
        from .driver import DriverInlet
        if world > 1:
            _refuse("&BC BCxm = 3: one GPU (the handle with open x boundaries is a one-rank handle)")
        inlet = DriverInlet(wdir, int(deck.get("DRIVER", "driverjobnr")), core.g.ny, core.g.nz, int(deck.get("DRIVER", "driverstore")),
                            thl=core.ltempeq and int(deck.get("BC", "BCxT")) == 3, qt=core.lmoist and int(deck.get("BC", "BCxq")) == 3,
                            nsv=core.nsv if int(deck.get("BC", "BCxs")) == 3 else 0)
        inlet.hand_over(core, timee)      # (the start-up's `boundary` applies them: rk3step = 0, src/modboundary.f90:262-266)
    forcings = LevelForcings(core, deck)
    if warm < 0:      # the reference's order at a cold start: thermodynamics on the fields as read (src/modstartup.f90:1601), then boundary
        core.start_up(before_boundary=forcings.capture_startup, dtmax=float(deck.get("RUN", "dtmax")))
    tdump = None
    for sw in ("lydump", "lxydump", "ltkedump", "lkslicedump", "lislicedump", "ljslicedump"):
        if deck.is_set("OUTPUT", sw) and deck.get("OUTPUT", sw):
            sys.stderr.write(f" WARNING: &OUTPUT {sw}: this dump is not written by the device runner (tdump, xytdump, mintdump, fielddump are)\n")
    fdump = None
    if bool(deck.get("OUTPUT", "lfielddump")):
        fdump = FieldDump(core, deck.get("OUTPUT", "tfielddump"), deck.get("OUTPUT", "fieldvars"), wdir, iexp, rank)
    if any(bool(deck.get("OUTPUT", sw)) for sw in ("ltdump", "lxytdump", "lmintdump", "lytdump")):      # src/modstatsdump.f90
        from .stats import TDump
        lists = None
        if deck.get("RUN", "libm"):
            from .ibm import read_ibm
            lists = read_ibm(deck)
        tdump = TDump(core, float(deck.get("OUTPUT", "tsample")), float(deck.get("OUTPUT", "tstatsdump")),
                      float(deck.get("OUTPUT", "tstatstart")), wdir=wdir, expnr=iexp, xyt=bool(deck.get("OUTPUT", "lxytdump")),
                      ibm_lists=lists, wrap=(int(deck.get("RUN", "nprocx")) > 1, int(deck.get("RUN", "nprocy")) > 1),
                      jtot=int(deck.get("DOMAIN", "jtot")), j0=rank * nyl, nyl=nyl, yt=bool(deck.get("OUTPUT", "lytdump")))
        tdump.mint = bool(deck.get("OUTPUT", "lmintdump"))
    # (the reference restarts the restart clock and the step counter on a warm start: tnextrestart = trestart,
    # ntrun = 0, src/modglobal.f90:869; this runner keeps counting from the file it started from, so that the files
    # of a continued run do not overwrite those of the first leg)
    tnext = timee + trestart
    t_end = timee + runtime
    say = (lambda *a: None) if (args.quiet or rank) else (lambda *a: print(*a, flush=True))
    say(f"udcore.run: {core.g.nx}x{core.g.ny}x{core.g.nz} cells on {world} GPU(s), dtmax = {dtmax}, runtime = {runtime}")
    t0, nsteps = time.perf_counter(), 0
    while core.timee < t_end - 1e-12 and (args.steps <= 0 or nsteps < args.steps):
        for _ in range(3):                                     # one full RK3 step
            rk, dt = core.tstep_update(dtmax, ladaptive, courant, diffnr)
            forcings.update(rk, dt)
            if inlet is not None and rk == 3 and not core.timee > t_end:      # drivergen of stage 3's `boundary` (inside the fused substep);
                inlet.hand_over(core, core.timee)                             # "if (timee > runtime + btime) return", src/moddriver.f90:216
            core.substep(rk, dt, with_forces=True)
            if tdump is not None and tdump.step(rk, dt, core.timee) == "dump":
                say(f"  tdump written at timee = {core.timee:.6f} ({tdump.nsamples} samples so far)")
            if tdump is not None and core.open_x and rk == 3:
                # (inflow / outflow with the statistics on: stage 3 of the fused substep ends ahead of `boundary`, where the reference samples,
                #  src/program.f90:199-214; udc_boundary and udc_thermodynamics follow the sample)
                core.boundary()
                if core.moist_thermo:
                    core.thermodynamics()
        nsteps += 1
        ntrun += 1
        if fdump is not None and fdump.step(core.timee):
            say(f"  fielddump written at timee = {core.timee:.6f}")
        if core.timee >= tnext:                                # writerestartfiles, src/modsave.f90:77
            tnext += trestart
            R.save_restart(core, wdir, iexp, ntrun, core.timee, core.dt, rank=rank, fill={"thl0": deck.thl[0]})
            say(f"  restart files written at ntrun = {ntrun}, timee = {core.timee:.6f}")
        if nsteps % 50 == 0:
            divmax, _ = core.divergence()
            say(f"  step {nsteps}: timee = {core.timee:.4f} dt = {core.dt:.5f} divmax = {divmax:.2e}")
    core.sync()
    wall = time.perf_counter() - t0
    divmax, _ = core.divergence()
    paths = R.save_restart(core, wdir, iexp, ntrun, core.timee, core.dt, rank=rank, fill={"thl0": deck.thl[0]})
    cells = core.g.nx * core.g.ny * core.g.nz
    say(f"udcore.run: {nsteps} steps to timee = {core.timee:.6f} in {wall:.2f} s "
        f"({cells * 3 * nsteps / max(wall, 1e-9):.3e} cell-updates/s), divmax = {divmax:.2e}; restart: {os.path.basename(paths[0])}")
    if at_end is not None:
        at_end(core, tdump)
    core.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
