"""One GPU: the forward half of the Poisson solve in udc_fft.hip (fillps' divergence inside the x transform, y pass in registers:
ny = 128, 256, 512; UDC_OWN_FWD=1 also the Stockham y pass of the other power-of-two sizes) against div_rhs + rocFFT's forward plan
on the same state (src/modpois.f90:968-970 fillps, :459-702 the transforms), and against the CPU oracle at the smallest size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(g, seed):
    import oracle_lib as ol
    rng = np.random.default_rng(seed)
    o = ol.Oracle(g.nx, g.ny, g.nz, g.dx, g.dy, g.dzf, g.dzh, sgs=2, csz=0.21658244510412, lbottom=True, z0=0.05)
    st = {}
    for k in ("u0", "v0", "w0"):
        a = np.zeros(g.mshape())
        a[1:-1, 1:-1, 1:-1] = (1.0 if k == "u0" else 0.0) + 0.05 * rng.standard_normal((g.nz, g.ny, g.nx))
        st[k] = a
    st["w0"][1] = 0.
    for k in ("u0", "v0", "w0"):
        o.call("orc_halos_m", st[k])
    for k, m in (("u0", "um"), ("v0", "vm"), ("w0", "wm")):
        st[m] = st[k].copy()
    for k in ("up", "vp", "wp", "pres0", "ekm", "ekh", "p", "pup", "pvp", "pwp"):
        st[k] = np.zeros(g.mshape())
    o.call("orc_boundary", st["u0"], st["v0"], st["w0"], st["um"], st["vm"], st["wm"], None, None)
    return o, st


def _run(g, st, nsub, dt):
    from udcore.core import DynCore
    from udcore import lib as L
    core = DynCore(g, sgs=L.SGS_VREMAN, lbottom=True, z0=0.05)
    core.load_state({k: st[k] for k in ("u0", "v0", "w0", "um", "vm", "wm", "pres0")})
    for isub in range(nsub):
        core.substep(isub % 3 + 1, dt, with_forces=False)
    out = {k: core.download(k) for k in ("u0", "v0", "w0", "pres0")}
    divmax, _ = core.divergence()
    core.close()
    return out, divmax


@pytest.mark.parametrize("shape,force", [((32, 128, 8), None), ((64, 256, 6), None), ((16, 512, 4), None), ((128, 256, 5), None),
                                         ((32, 64, 8), "1"), ((64, 32, 6), "1")])
def test_own_forward_equals_rocfft_forward(shape, force, monkeypatch):
    from udcore.grid import Grid
    nx, ny, nz = shape
    g = Grid.uniform(nx, ny, nz)
    o, st = _state(g, 7)
    dt = 0.1
    monkeypatch.setenv("UDC_OWN_FWD", "0")
    ref, div_ref = _run(g, st, 4, dt)
    if force:
        monkeypatch.setenv("UDC_OWN_FWD", force)
    else:
        monkeypatch.delenv("UDC_OWN_FWD")           # the default takes the own forward half at these sizes
    own, div_own = _run(g, st, 4, dt)
    for k in ref:
        scale = max(np.abs(ref[k]).max(), 1e-30)
        err = np.abs(own[k][1:-1, 1:-1, 1:-1] - ref[k][1:-1, 1:-1, 1:-1]).max() / scale
        # two FFT algorithms, four substeps: round-off (the pressure carries the solve's conditioning; the run fixtures allow 1e-9)
        assert err <= (1e-10 if k == "pres0" else 1e-11), (shape, k, err)
    assert div_own < 1e-11 and div_ref < 1e-11, (div_own, div_ref)


def test_own_forward_against_oracle():
    from udcore.grid import Grid
    g = Grid.uniform(32, 128, 8)
    o, st = _state(g, 11)
    dt = 0.1
    out, divmax = _run(g, st, 1, dt)
    o.substep(st, 1, dt)
    for k in out:
        err = np.abs(out[k][1:-1, 1:-1, 1:-1] - st[k][1:-1, 1:-1, 1:-1]).max() / max(np.abs(st[k]).max(), 1e-30)
        assert err < 1e-10, (k, err)
    assert divmax < 1e-11


def test_switch_is_visible_in_the_profile(monkeypatch):
    """the default path at 32 x 128 x 8 launches the fused x kernel, not div_rhs (so the comparison above compares two paths)"""
    from udcore.grid import Grid
    from udcore.core import DynCore
    from udcore import lib as L
    g = Grid.uniform(32, 128, 8)
    o, st = _state(g, 3)
    names = {}
    for tag, env in (("own", None), ("rocfft", "0")):
        if env is None:
            monkeypatch.delenv("UDC_OWN_FWD", raising=False)
        else:
            monkeypatch.setenv("UDC_OWN_FWD", env)
        core = DynCore(g, sgs=L.SGS_VREMAN, lbottom=True, z0=0.05)
        core.load_state({k: st[k] for k in ("u0", "v0", "w0", "um", "vm", "wm", "pres0")})
        core.profile(True)
        core.substep(1, 0.1, with_forces=False)
        core.sync()
        names[tag] = set(core.profile_get())
        core.close()
    assert "fftx_pack_fwd" in names["own"] and "div_rhs" not in names["own"], names["own"]
    assert "div_rhs" in names["rocfft"] and "fftx_pack_fwd" not in names["rocfft"], names["rocfft"]


@pytest.mark.parametrize("shape", [(16, 8, 4), (32, 16, 5), (64, 8, 4), (128, 16, 6), (256, 32, 4), (512, 16, 4), (1024, 8, 4), (2048, 8, 3)])
def test_slab_line_transforms_of_every_length(shape, monkeypatch):
    """The slab ranks' own x transforms at every line length they accept (nx = 16 .. 2048: the backward one as radix-8 butterflies in
    registers, 2^c 8^a complex points for every c in 0, 1, 2 and a in 1 .. 3; mirrored Thomas pairs where a line has 16 rows) against the
    single-slab path (2-D rocFFT) on the same state: src/modpois.f90:459-702."""
    from udcore.grid import Grid
    nx, ny, nz = shape
    g = Grid.uniform(nx, ny, nz)
    o, st = _state(g, 5)
    dt = 0.1
    monkeypatch.setenv("UDC_OWN_FWD", "0")
    ref, div_ref = _run(g, st, 3, dt)
    monkeypatch.setenv("UDC_FORCE_SLAB", "1")
    monkeypatch.setenv("UDC_THOMAS_MIRROR_MIN", "16")
    own, div_own = _run(g, st, 3, dt)
    for k in ref:
        scale = max(np.abs(ref[k]).max(), 1e-30)
        err = np.abs(own[k][1:-1, 1:-1, 1:-1] - ref[k][1:-1, 1:-1, 1:-1]).max() / scale
        assert err <= (1e-10 if k == "pres0" else 1e-11), (shape, k, err)
    assert div_own < 1e-11 and div_ref < 1e-11, (div_own, div_ref)
