"""Stand-in for ``proxmin.algorithms`` (third party, absent from the container).

Only the AMSGrad moments are provided, and they are OUR restatement
(``oracle.pgm.amsgrad_phi_psi``), not proxmin code: goldens produced through this
function pin the reference's own ``AdaproxParameter.update`` / ``LiteBlend.fit`` logic
around it, not these five lines (see oracle/__init__.py, "parity unpinned")."""


def _missing(*args, **kwargs):
    raise NotImplementedError("proxmin is not available in this container")


def _amsgrad_phi_psi(it, G, M, V, Vhat, b1, b2, eps, p):
    from oracle.pgm import amsgrad_phi_psi

    return amsgrad_phi_psi(it, G, M, V, Vhat, b1[it], b2, eps)


_adam_phi_psi = _nadam_phi_psi = _missing
_padam_phi_psi = _adamx_phi_psi = _radam_phi_psi = _missing
