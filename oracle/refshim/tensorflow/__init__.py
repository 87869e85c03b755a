"""NumPy-backed stand-in for the handful of ``tf.*`` primitives that the
reference's ``gan_utils.py`` touches (``/root/reference/gan_utils.py:3``).

TEST INFRASTRUCTURE ONLY.  TensorFlow is not installed in the build container
(``import gan_utils`` raises ``ModuleNotFoundError: tensorflow`` -- an ordinary
Python error).  Putting this directory first on
``sys.path`` lets ``tests/golden/make_golden.py`` import the reference file
*verbatim from /root/reference* and execute its own control flow and op order
(positional-argument quirk, Lmin=100 stop rule, u-then-v update order, ...)
to produce the golden vectors under ``tests/golden/``.

What is pinned by this route: the reference's algorithm as written.
What is NOT pinned: TensorFlow/Eigen's floating-point summation order -- the
primitives below are NumPy's (pairwise ``np.sum``), so agreement with real TF
is to fp32 rounding, not bitwise.  Nothing here is imported by the product
package, and nothing from /root/reference is copied into the repo.

``float32`` is a module attribute so the generator can re-run the identical
reference code in float64 (``set_float(np.float64)``) to get a high-precision
value of the same algorithm.
"""
import numpy as _np

__version__ = "0.0-numpy-standin"

float32 = _np.float32
float64 = _np.float64
newaxis = None

# number of reduce_logsumexp calls so far: the reference does not return its
# executed Sinkhorn iteration count (gan_utils.py:148,158 keep it local), but it
# makes exactly two LSE calls per iteration, so the generator reads it off here.
lse_calls = 0


def set_float(dtype):
    """Re-point ``tf.float32`` (used by the reference for every cast and
    constant) at ``dtype``; float64 gives the high-precision golden values."""
    global float32
    float32 = dtype


def expand_dims(x, axis):
    return _np.expand_dims(_np.asarray(x), axis)


def reduce_sum(x, axis=None, keepdims=False):
    return _np.sum(_np.asarray(x), axis=axis, keepdims=keepdims)


def reduce_max(x, axis=None, keepdims=False):
    return _np.max(_np.asarray(x), axis=axis, keepdims=keepdims)


def transpose(x, perm=None):
    return _np.transpose(_np.asarray(x), perm)


def cast(x, dtype):
    return _np.asarray(x).astype(dtype)[()]


def shape(x):
    return _np.asarray(x).shape


def ones(shape, dtype=None):
    if dtype is None:
        dtype = float32
    return _np.ones(shape, dtype=dtype)


def squeeze(x, axis=None):
    return _np.squeeze(_np.asarray(x), axis=axis)


def reshape(x, shape):
    return _np.reshape(_np.asarray(x), [int(s) for s in shape])


def exp(x):
    return _np.exp(x)


def range(*args, **kwargs):  # noqa: A001 - mirrors tf.range
    kwargs.pop("dtype", None)
    return _np.arange(*args)


def reduce_logsumexp(x, axis=None, keepdims=False):
    """TF semantics: max-shifted, the max treated as a constant and replaced by
    0 where it is not finite (tensorflow/python/ops/math_ops.py
    ``reduce_logsumexp``)."""
    global lse_calls
    lse_calls += 1
    x = _np.asarray(x)
    raw_max = _np.max(x, axis=axis, keepdims=True)
    my_max = _np.where(_np.isfinite(raw_max), raw_max, _np.zeros_like(raw_max))
    out = _np.log(_np.sum(_np.exp(x - my_max), axis=axis, keepdims=True)) + my_max
    if not keepdims:
        out = _np.squeeze(out, axis=axis)
    return out


class _Math:
    @staticmethod
    def log(x):
        return _np.log(x)

    @staticmethod
    def abs(x):
        return _np.abs(x)

    @staticmethod
    def add(x, y):
        return _np.add(x, y)

    @staticmethod
    def subtract(x, y):
        return _np.subtract(x, y)

    @staticmethod
    def greater(x, y):
        return bool(_np.all(_np.greater(x, y)))

    reduce_logsumexp = staticmethod(reduce_logsumexp)

    @staticmethod
    def reduce_std(x, axis=None, keepdims=False):
        # population standard deviation (ddof = 0), as tf.math.reduce_std
        x = _np.asarray(x)
        mean = _np.mean(x, axis=axis, keepdims=True)
        var = _np.mean(_np.square(x - mean), axis=axis, keepdims=keepdims)
        return _np.sqrt(var)


math = _Math()
