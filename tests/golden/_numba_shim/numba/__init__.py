"""Identity-decorator stand-in for numba, used ONLY by tests/golden/gen_golden.py.

The reference imports `njit`/`jit` at module import; every use is `cache=True`
only (no fastmath / parallel), so replacing them by identity decorators leaves
the IEEE semantics of the reference untouched.
"""


def _ident(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit = _ident
