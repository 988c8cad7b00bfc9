def _missing(*args, **kwargs):
    raise NotImplementedError("proxmin is not available in this container")


_adam_phi_psi = _nadam_phi_psi = _amsgrad_phi_psi = _missing
_padam_phi_psi = _adamx_phi_psi = _radam_phi_psi = _missing
