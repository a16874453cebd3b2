"""Deterministic synthetic inputs for the atmosphere -> TOD path.

Real runs take their pointing, detector tables, weather and ``am`` emission
tables from maria's front end; those need network data that is not available
here, so the benchmark and the tests build inputs of the same shape from closed
forms.  Everything here is host-side numpy set-up, not the hot path.
"""

from __future__ import annotations

import numpy as np

SEED = 20260612

# reference layer midpoints, atmosphere/extrusion.py:66,79
LAYER_HEIGHTS = np.array([250.0, 750.0, 1250.0, 1750.0, 2500.0, 4000.0, 6500.0, 10000.0])

# named workloads of BASELINE.json["configs"]
CONFIGS = {
    # name: (D, n_bands, fov_deg, fs_Hz, duration_s, n_layers, screen_side)
    "mustang2_60s": dict(n_det=217, n_bands=1, fov_deg=0.07, fs=50.0, duration=60.0, n_layers=1, side=256),
    "mustang2_600s": dict(n_det=217, n_bands=1, fov_deg=0.07, fs=50.0, duration=600.0, n_layers=4, side=1024),
    "act_3k": dict(n_det=9000, n_bands=3, fov_deg=1.0, fs=400.0, duration=600.0, n_layers=8, side=2048),
    "atlast_10k": dict(n_det=10000, n_bands=1, fov_deg=2.0, fs=400.0, duration=600.0, n_layers=8, side=2048),
    "atlast_50k": dict(n_det=50000, n_bands=1, fov_deg=2.0, fs=400.0, duration=3600.0, n_layers=16, side=4096),
}


def layer_heights(n_layers: int) -> np.ndarray:
    """First ``n_layers`` reference midpoints, extended geometrically above 10 km."""
    h = list(LAYER_HEIGHTS[:n_layers])
    while len(h) < n_layers:
        h.append(h[-1] * 1.15)
    return np.array(h)


def hex_pack(n: int, fov_rad: float) -> np.ndarray:
    """``n`` focal-plane offsets (radians) on a hexagonal lattice filling ``fov_rad``,
    ordered ring by ring so neighbours in memory are neighbours on the sky."""
    if n == 1:
        return np.zeros((1, 2))
    rings = int(np.ceil((np.sqrt(12 * n - 3) - 3) / 6)) + 1
    pts = []
    for q in range(-rings, rings + 1):
        for r in range(max(-rings, -q - rings), min(rings, -q + rings) + 1):
            pts.append((q + r / 2.0, r * np.sqrt(3) / 2.0))
    pts = np.array(pts)
    rad = np.hypot(pts[:, 0], pts[:, 1])
    ang = np.arctan2(pts[:, 1], pts[:, 0])
    order = np.lexsort((ang, np.round(rad, 9)))
    pts = pts[order][:n]
    scale = 0.5 * fov_rad / max(np.hypot(pts[:, 0], pts[:, 1]).max(), 1e-30)
    return pts * scale


def _daisy_curve(phase, a, b, petals, miss_freq):
    """The daisy's closed form (plan/patterns.py:108-113): a rose of ``petals`` lobes of
    amplitude ``a`` plus a slow ``b`` term that makes successive petals miss the centre,
    rescaled so that the farthest point lies at ``a + b``."""
    lobe, swing = np.cos(petals * phase), np.sin(petals * phase)
    x = a * lobe * np.sin(phase) + b * swing * np.cos(miss_freq * phase)
    y = a * lobe * np.cos(phase) + b * swing * np.sin(miss_freq * phase)
    return (a + b) * np.stack([x, y]) / np.sqrt((x * x + y * y).max())


def daisy_offsets(time, x_throw, y_throw=None, speed=0.5, petals=np.sqrt(np.e), miss_factor=0.2, miss_freq=0.1):
    """Boresight offsets [2, T] of the reference's ``daisy`` scan pattern (plan/patterns.py:116-155),
    in the units of ``x_throw``: the phase advances at ``speed / radius``, then up to four passes
    rescale the phase rate until the fastest sample moves at ``speed`` (within 1 %)."""
    time = np.asarray(time, float)
    y_throw = x_throw if y_throw is None else y_throw
    radius = x_throw
    if not radius > 0 or len(time) == 0:
        return np.zeros((2, len(time)))
    a = radius / (1 + miss_factor)
    b = a * miss_factor
    if len(time) < 2:  # a single sample has no speed to normalise (the reference's np.gradient raises)
        return _daisy_curve(np.zeros(len(time)), a, b, petals, miss_freq) * np.array([[1.0], [y_throw / x_throw]])
    dt = np.gradient(time)
    dphase = (speed / radius) * dt
    for _ in range(4):
        phase = np.cumsum(dphase)
        tx, ty = _daisy_curve(phase, a, b, petals, miss_freq)
        fastest = np.sqrt((np.gradient(tx) / dt) ** 2 + (np.gradient(ty) / dt) ** 2).max()
        if abs(np.log(fastest / speed)) <= 0.01:
            break
        dphase = dphase * (speed / fastest)
    x, y = _daisy_curve(phase, a, b, petals, miss_freq)
    return np.stack([x, (y_throw / x_throw) * y])


def offsets_to_phi_theta(dx, dy, c_phi, c_theta):
    """coords/transforms.py:10-29 in float32 (jax's default precision): the direction at
    offset (dx, dy) radians from the centre (c_phi, c_theta).  Host-side set-up only; the
    kernels carry their own copy of this chain (csrc/mrx_sample.hip)."""
    f32 = np.float32
    dx, dy = np.asarray(dx, f32), np.asarray(dy, f32)
    r = np.sqrt(dx * dx + dy * dy)
    p = np.arctan2(-dx, -dy)
    a = np.asarray(c_theta, f32) - f32(np.pi / 2)
    a_re, a_im = np.sin(r) * np.cos(p), np.cos(r)
    re = a_re * np.cos(a) - a_im * np.sin(a)
    im = a_re * np.sin(a) + a_im * np.cos(a)
    return np.arctan2(np.sin(r) * np.sin(p), re) + np.asarray(c_phi, f32), np.arcsin(im)


def daisy_scan(t, radius_deg=0.5, speed_deg_s=0.5, az_deg=45.0, el_deg=60.0, petals=np.sqrt(np.e)):
    """Boresight (az, el in radians) of ``Plan.generate(scan_pattern="daisy", frame="az/el")``:
    the daisy offsets about the scan centre (plan/plan.py:90-140)."""
    off = np.radians(daisy_offsets(t, radius_deg, radius_deg, speed_deg_s, petals=petals))
    az, el = offsets_to_phi_theta(off[0], off[1], np.radians(az_deg), np.radians(el_deg))
    return az.astype(np.float64), el.astype(np.float64)


def coarse_grid(t, az, el, timestep):
    """coordinates.py:286-304: ``arange(t.min, t.max, timestep)`` + linear interpolation
    (t is ascending, so np.interp equals interp1d inside the range; nothing is
    extrapolated because the coarse grid never leaves [t.min, t.max))."""
    ta = np.arange(t.min(), t.max(), timestep)
    return ta, np.interp(ta, t, az), np.interp(ta, t, el)


def emission_tables(n_bands: int, n_pwv: int = 32, n_el: int = 32):
    """Smooth stand-ins for the band-integrated ``am`` tables of band/band.py:272-280:
    P = a (T/270) (1 - exp(-(tau0 + kappa pwv)/sin el)) on a (T, pwv, el) grid whose
    last elevation node is 90.1 deg as in spectrum/atmosphere.py:48-50."""
    T = np.array([250.0, 270.0, 290.0])
    pwv = np.linspace(0.0, 10.0, n_pwv)
    el = np.radians(np.linspace(10.0, 90.0, n_el))
    el[-1] = np.radians(90.1)
    params = [(20.0, 0.03, 0.010), (30.0, 0.04, 0.025), (40.0, 0.06, 0.060), (55.0, 0.10, 0.150)]
    tables = []
    for b in range(n_bands):
        a, tau0, kappa = params[b % len(params)]
        tau = (tau0 + kappa * pwv[None, :, None]) / np.sin(np.minimum(el, np.pi / 2))[None, None, :]
        values = a * (T[:, None, None] / 270.0) * (1.0 - np.exp(-tau))
        tables.append({"T": T, "pwv": pwv, "el": el, "values": values})
    return tables


def pwv_rms_profile(h, pwv0, pwv_rms_frac=0.03):
    """extrusion.py:96-105."""
    rel_var = (np.exp(-h / 1e3) * h ** (1.0 / 7.0)) ** 2
    return np.sqrt((pwv0 * pwv_rms_frac) ** 2 * rel_var / rel_var.sum())


def _pointing_f64(offsets, az, el):
    """float64 pointing (same formula as coords/transforms.py:10-29), used only to
    size the screens."""
    dx, dy = offsets[:, 0][:, None], offsets[:, 1][:, None]
    r = np.hypot(dx, dy)
    p = np.arctan2(-dx, -dy)
    z = (np.sin(r) * np.cos(p) + 1j * np.cos(r)) * np.exp(1j * (el[None, :] - np.pi / 2))
    return np.arctan2(np.sin(r) * np.sin(p), z.real) + az[None, :], np.arcsin(z.imag)


def make_problem(
    n_det=217,
    n_bands=1,
    fov_deg=0.07,
    fs=50.0,
    duration=60.0,
    n_layers=1,
    side=256,
    timestep=0.1,
    pwv0=1.0,
    t0=0.0,
    min_res=5.0,
    seed=SEED,
    T0=273.15,
    gain=False,
):
    """Geometry + tables of one observation; the screens' ``values`` are left
    ``None`` (fill them with :meth:`maria_amd.pipeline.DevicePath.generate_screens`
    on the GPU, or with ``oracle.screens.numpy_screen`` in CPU-only tests)."""
    rng = np.random.default_rng(seed)
    n_t = int(round(duration * fs))
    t = t0 + np.arange(n_t) / fs
    az, el = daisy_scan(t)
    ta, az_a, el_a = coarse_grid(t, az, el, timestep)
    offsets = hex_pack(n_det, np.radians(fov_deg))
    band_index = (np.arange(n_det) * n_bands // n_det).astype(np.int32)  # contiguous band blocks
    m00 = np.ones(n_det, np.float32)

    heights = layer_heights(n_layers)
    rms = pwv_rms_profile(heights, pwv0)

    # points that bound the footprint: boresight plus a ring at the FOV edge
    ring = 0.5 * np.radians(fov_deg) * 1.02 * np.c_[np.cos(np.linspace(0, 2 * np.pi, 24, endpoint=False)), np.sin(np.linspace(0, 2 * np.pi, 24, endpoint=False))]
    phi, theta = _pointing_f64(np.r_[np.zeros((1, 2)), ring], az_a, el_a)
    px, py = np.cos(phi) / np.tan(theta), np.sin(phi) / np.tan(theta)

    layers = []
    for l, h in enumerate(heights):
        speed = 10.0 + 2.0 * l
        direction = np.radians(30.0 + 20.0 * l)
        vx = speed * np.cos(direction) * np.ones(len(ta))
        vy = speed * np.sin(direction) * np.ones(len(ta))
        # extrusion axis along the wind, like the reference's aligning transform
        ce, se = np.cos(direction), np.sin(direction)
        transform = np.array([[ce, -se, 0.0], [se, ce, 0.0], [0.0, 0.0, 1.0]])
        tx, ty = np.cumsum(timestep * vx), np.cumsum(timestep * vy)
        e = (h * px + tx[None]) * ce + (h * py + ty[None]) * se
        c = -(h * px + tx[None]) * se + (h * py + ty[None]) * ce
        extent = max(np.ptp(e), np.ptp(c))
        res = max(min_res, extent * 1.02 / (side - 8))
        # start + i*step, the form np.arange / np.linspace produce (atmosphere.py:208-245)
        axis_e = (0.5 * (e.min() + e.max()) - 0.5 * res * (side - 1)) + res * np.arange(side)
        axis_c = (0.5 * (c.min() + c.max()) - 0.5 * res * (side - 1)) + res * np.arange(side)
        layers.append(
            dict(
                h=float(h),
                pwv_rms=float(rms[l]),
                vx=vx,
                vy=vy,
                transform=transform,
                extrusion=axis_e,
                cross_section=axis_c,
                res=float(res),
                r0=float(max(1e3, 300.0 + h / 10.0)),  # atmosphere.py:247
                nu=5.0 / 6.0,
                beam_sigma=50.0 / 2.355,  # metres; SURVEY 8(d)
                values=None,
            )
        )

    problem = dict(
        t=t,
        ta=ta,
        az_a=az_a,
        el_a=el_a,
        offsets=offsets,
        band_index=band_index,
        m00=m00,
        layers=layers,
        tables=emission_tables(n_bands),
        T0=float(T0),
        pwv0=float(pwv0),
        timestep=float(timestep),
        gain=np.exp(0.05 * rng.standard_normal(n_det)).astype(np.float32) if gain else None,
        seed=int(seed),
        fs=float(fs),
    )
    return problem


def config_problem(name: str, **overrides):
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    return make_problem(**kw)
