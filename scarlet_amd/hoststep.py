"""Plug-in seam of the fit (reference constraint.py:39-55, blend.py:135-138): parameters
whose proximal operator or step rule is user code.

``Blend.fit`` hands every parameter's constraint to ``proxmin.adaprox`` as a Python
callable ``prox(X, step)`` and evaluates callable steps as ``step(X, it)``.  The device
loop only knows the built-in chains (``constraint.device_flags``).  A parameter with
anything else -- a ``Prior``, a ``Constraint`` subclass, a built-in chain in another order,
``MonotonicMaskConstraint``, a custom step callable -- stays with the
host: per iteration the device still renders, convolves and gathers the gradient of
every parameter (``smi_batch_gradient``) and updates all parameters it can express;
the host takes the AMSGrad step and the proximal sub-iterations of the rest with the
user's callables and uploads the result.

The arithmetic here restates the device kernels (``csrc/kernels.hip``:
``update_spectrum``, ``update_kernel_reg``) operation by operation in float32,
including the order of the wavefront reductions, so that a user constraint that does
what a built-in one does gives bit-identical results to the device path.
"""

import numpy as np

F32 = np.float32
SCHEMES = ("adam", "nadam", "amsgrad", "padam", "adamx", "radam")


def wave_sum(x):
    """Sum of a float32 array the way the 64-lane reductions of the kernels form it:
    lane l adds its elements l, l + 64, ... in ascending order, then the lanes are
    combined in a balanced binary tree (DPP quad swaps, half-row and row mirrors,
    readlanes of the four rows)."""
    x = np.asarray(x, dtype=F32).reshape(-1)
    n = -(-x.size // 64) * 64
    lanes = np.zeros(n, dtype=F32)
    lanes[: x.size] = x
    rows = lanes.reshape(-1, 64)
    acc = np.zeros(64, dtype=F32)
    for row in rows:
        acc = acc + row
    while acc.size > 1:
        acc = acc[0::2] + acc[1::2]
    return F32(acc[0])


def _max_nan(a, floor):
    """np.maximum (a NaN in the data propagates), as the kernels' max_nan."""
    return np.maximum(a, F32(floor))


class HostParameter:
    """One ``Parameter`` of a factorized component that the host updates.

    kind: "sed" (spectrum, (C,)) or "morph" (image, (h, w))
    step: ``(constant, relative factor, minimum)`` of a built-in rule, or the user's
        callable ``step(X, it)``
    """

    def __init__(self, parameter, kind, step, scheme="amsgrad", p=0.25):
        self.p = parameter
        self.kind = kind
        self.step = step
        if scheme not in SCHEMES:
            raise ValueError("scheme must be one of {}".format(SCHEMES))
        if scheme != "amsgrad":
            import warnings

            # proxmin (the reference's optimizer package) is not available to pin these against
            warnings.warn("scheme={!r} follows the published algorithm, not a check against "
                          "proxmin.adaprox: parity unpinned".format(scheme), stacklevel=3)
        self.scheme = scheme
        self.padam_p = p
        for name in ("m", "v", "vhat"):
            value = getattr(parameter, name)
            setattr(self, name, np.zeros(parameter.shape, F32) if value is None
                    else np.array(value, dtype=F32))

    def alpha(self, it):
        x = np.asarray(self.p, dtype=F32)
        if callable(self.step):
            return np.asarray(self.step(self.p, it=it), dtype=F32)
        const, rel, minimum = self.step
        mean = wave_sum(x) / F32(x.size)
        if self.kind == "sed":  # update_spectrum: fmaxf(min_step_c, rel * mean)
            floor = np.maximum(np.asarray(minimum, dtype=np.float64), const).astype(F32)
            return np.maximum(np.broadcast_to(floor, x.shape), F32(rel) * mean)
        # update_kernel_reg: fmaxf(morph_step, rel * (sum / N))
        return np.maximum(F32(max(const, float(np.max(minimum)))), F32(rel) * mean)

    def update(self, it, g, e_rel, prox_max_iter, b1, b2, eps):
        """AMSGrad step (a tenth of it at ``it == 0``) and proximal sub-iterations
        (lite/parameters.py:274-305), in place on the Parameter."""
        p = self.p
        x0 = np.array(p, dtype=F32)
        g = np.zeros_like(x0) if p.fixed else np.asarray(g, dtype=F32).reshape(x0.shape)
        if p.prior is not None and not p.fixed:
            # the reference adds what the prior returns for the current value to the
            # likelihood's gradient (blend.py:120-131: ``x.prior(x)``, not ``prior.grad``)
            g = g + np.asarray(p.prior(np.asarray(p).view(np.ndarray)), dtype=F32)
        b1, b2, eps, one = F32(b1), F32(b2), F32(eps), F32(1)
        alpha = self.alpha(it)  # on the pre-update values (blend.py:135-138)
        self.m = (one - b1) * g + b1 * self.m
        self.v = (one - b2) * g * g + b2 * self.v
        phi, psi = self.phi_psi(it, g, b1, b2, eps)
        upd = alpha * phi / psi
        if it == 0:
            upd = upd / F32(10)
        x = x0 - upd
        prox = p.constraint
        z = x
        if prox is not None:
            pmax = F32(psi.max())
            # the spectrum kernel divides, the image kernel multiplies by the reciprocal
            ratio = psi / pmax if self.kind == "sed" else psi * (one / pmax)
            gamma = alpha / pmax
            e2 = F32(e_rel) * F32(e_rel)
            z = x.copy()
            for _ in range(prox_max_iter):
                zn = np.asarray(prox(z - ratio * (z - x), gamma), dtype=F32)
                d2 = wave_sum((zn - z) * (zn - z))
                z2 = wave_sum(z * z)
                z = zn
                if d2 <= e2 * z2:
                    break
        p[...] = z
        return z

    def phi_psi(self, it, g, b1, b2, eps):
        """Direction ``phi`` and metric ``psi`` of the step ``x -= alpha phi / psi`` from the
        moments already updated with ``g`` (``proxmin.adaprox(scheme=...)``, blend.py:144;
        the reference lists the schemes at lite/parameters.py:158-165).  "amsgrad" is the
        device kernels' arithmetic.  proxmin is not available here: the other schemes
        follow the papers the reference cites (lite/parameters.py:187-193) -- parity
        unpinned."""
        one, t = F32(1), it + 1
        if self.scheme in ("amsgrad", "adamx", "padam"):
            # AdamX rescales vhat by ((1 - b1_t) / (1 - b1_{t-1}))^2, which is 1 for the
            # constant b1 of Blend.fit (Phuong & Phong 2019)
            self.vhat = self.v.copy() if it == 0 else np.maximum(self.vhat, self.v)
            vh = np.maximum(self.vhat, eps)
            if self.scheme == "padam":  # Chen & Gu 2018: vhat^p instead of the square root
                return self.m, np.power(vh, F32(self.padam_p))
            return self.m, np.sqrt(vh)
        c1, c2 = one - F32(b1) ** t, one - F32(b2) ** t  # bias corrections
        if self.scheme == "adam":  # Kingma & Ba 2015
            return self.m / c1, np.sqrt(self.v / c2) + eps
        if self.scheme == "nadam":  # Dozat 2016: Nesterov look-ahead on the first moment
            return (b1 * self.m + (one - b1) * g) / c1, np.sqrt(self.v / c2) + eps
        # RAdam (Liu et al. 2019): variance rectification once rho_t > 4, else plain momentum
        rho_inf = 2.0 / (1.0 - float(b2)) - 1.0
        rho = rho_inf - 2.0 * t * float(b2) ** t / (1.0 - float(b2) ** t)
        if rho > 4:
            r = np.sqrt((rho - 4) * (rho - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho))
            psi = np.maximum(np.sqrt(self.v / c2) / F32(r), np.sqrt(eps))
        else:
            psi = np.ones_like(self.v)
        return self.m / c1, psi

    def store(self):
        """Leave the moments on the Parameter like the device path does (float64)."""
        self.p.m, self.p.v, self.p.vhat = (a.astype(np.float64) for a in (self.m, self.v, self.vhat))


class HostVector:
    """A free 2-vector -- the sub-pixel shift of an image, the centre of a point source, a
    renderer's ``psf_shift`` -- that carries a ``Prior``, a constraint or a step callable:
    the reference treats it like every other ``Parameter`` (blend.py:120-145), the device
    only knows the bare AMSGrad step of the defaults (morphology.py:673-676, source.py:115,
    renderer.py:175-177).  Such a vector keeps a device step of 0; the host takes its step
    from the device's gradient in float64 -- ``amsgrad_pair`` of csrc/shift.hip operation by
    operation, so a vector without any of the three moves exactly as on the device -- and
    writes the result back (``smi_batch_set_centers`` / the kernel of the observation).

    step: ``(constant, relative factor, minimum)`` of a built-in rule, or the user's callable
    """

    kind = "vec"

    def __init__(self, parameter, step):
        self.p = parameter
        self.step = step
        for name in ("m", "v", "vhat"):
            value = getattr(parameter, name)
            setattr(self, name, np.zeros(2) if value is None else np.array(value, dtype=np.float64))

    def alpha(self, it):
        if callable(self.step):
            return np.asarray(self.step(self.p, it=it), dtype=np.float64)
        const, rel, minimum = self.step
        x = np.asarray(self.p, dtype=np.float64)
        # (relative_step, parameter.py:126-129: max(minimum, factor * mean), on the pre-update value)
        return max(max(float(const), float(np.max(minimum))), float(rel) * 0.5 * float(x[0] + x[1]))

    def update(self, it, g, e_rel, prox_max_iter, b1, b2, eps):
        p = self.p
        x0 = np.array(p, dtype=np.float64)
        g = np.zeros(2) if p.fixed else np.asarray(g, dtype=np.float64).reshape(2)
        if p.prior is not None and not p.fixed:
            # what the prior returns for the current value joins the gradient (blend.py:120-131)
            g = g + np.asarray(p.prior(np.asarray(p).view(np.ndarray)), dtype=np.float64)
        alpha = self.alpha(it)
        self.m = (1.0 - b1) * g + b1 * self.m
        self.v = (1.0 - b2) * g * g + b2 * self.v
        self.vhat = self.v.copy() if it == 0 else np.maximum(self.vhat, self.v)
        psi = np.sqrt(np.maximum(self.vhat, eps))
        upd = alpha * self.m / psi
        if it == 0:
            upd = upd / 10.0
        x = x0 - upd
        z = x
        prox = p.constraint
        if prox is not None:
            pmax = psi.max()
            gamma = alpha / pmax
            z = x.copy()
            for _ in range(prox_max_iter):
                zn = np.asarray(prox(z - psi / pmax * (z - x), gamma), dtype=np.float64)
                done = np.sum((zn - z) ** 2) <= e_rel ** 2 * np.sum(z ** 2)
                z = zn
                if done:
                    break
        p[...] = z
        return z

    def store(self):
        self.p.m, self.p.v, self.p.vhat = self.m.copy(), self.v.copy(), self.vhat.copy()


class HostBandSource:
    """A ``PointSource`` on an ``ImagePSF`` model PSF that DIFFERS between the bands
    (source.py:92-128 takes any ``frame.psf``; the morphology is then a cube, one stamp per
    band Fourier-shifted to the centre, morphology.py:476-513).  A device component is a
    spectrum x ONE image, so the device fits such a source as C components -- band c's spectrum
    entry (the others zero) x band c's stamp under a Fourier shift -- with device steps of 0,
    and the host steps the source's two Parameters from the gathered gradients:

        d/d sed[c]   = g_sed of band component c, entry c
        d/d center   = sum over the band components of their shift gradients

    the spectrum with ``HostParameter`` (float32, the spectrum kernel's arithmetic), the centre
    with ``HostVector`` (float64, the device's ``amsgrad_pair``).
    """

    kind = "band"

    def __init__(self, sed, center, sed_step, center_step, box_center, n_bands):
        self.sed = HostParameter(sed, "sed", sed_step)
        self.vec = HostVector(center, center_step)
        self.box_center = np.asarray(box_center, dtype=np.float64)
        self.n_bands = int(n_bands)
        self.p = center  # (finite check of the fit loop)

    def update(self, it, g_sed_rows, g_vec_rows, e_rel, prox_max_iter, **opt):
        g_sed = np.array([g_sed_rows[c][c] for c in range(self.n_bands)], dtype=np.float32)
        g_vec = np.sum(np.asarray(g_vec_rows, dtype=np.float64), axis=0)
        sed = self.sed.update(it, g_sed, e_rel, prox_max_iter, **opt)
        center = self.vec.update(it, g_vec, e_rel, prox_max_iter, **opt)
        return sed, center - self.box_center

    def store(self):
        self.sed.store()
        self.vec.store()
