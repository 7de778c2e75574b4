"""Identity-decorator stand-in for numba, used ONLY by tests/golden/make_golden.py
inside the build container to import the (pure Python) reference and run its
hot-path bodies under CPython.  Never imported by the product or the tests."""


def _identity(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn

    return wrap


njit = jit = vectorize = guvectorize = generated_jit = _identity
float64 = int64 = int8 = boolean = None
__version__ = "0.0-shim"
