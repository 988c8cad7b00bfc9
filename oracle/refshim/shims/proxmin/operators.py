from oracle.proxops import (  # noqa: F401
    prox_plus,
    prox_soft,
    prox_hard,
    prox_hard_plus,
    prox_unity_plus,
)
