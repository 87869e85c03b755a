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


# --------------------------------------------------------------------------------------------
# Round 2: the primitives ``/root/reference/data_utils.py`` needs so that ``KernelSmoothing``
# (:478-586), ``WarmUp`` (:589-621) and ``exponential_decay_with_warmup`` (:624-633) can be run
# verbatim by tests/golden/make_golden_smoothing.py.  The convolutions are torch's CPU kernels
# (``torch.nn.functional.conv1d/2d/3d``, cross-correlation like TF's) -- an implementation that
# shares nothing with oracle/smoothing_np.py or the HIP kernels.
# --------------------------------------------------------------------------------------------
def _range_with_dtype(*args, **kwargs):
    dtype = kwargs.pop("dtype", None)
    out = _np.arange(*args)
    return out if dtype is None else out.astype(dtype)


range = _range_with_dtype  # noqa: A001 - gan_utils.py calls tf.range(L) (integers, no dtype)


def constant(value, dtype=None):
    return _np.asarray(value, dtype=dtype)


def pad(tensor, paddings, mode="CONSTANT"):
    """tf.pad: REFLECT mirrors without repeating the edge sample (NumPy 'reflect'); SYMMETRIC
    repeats it (NumPy 'symmetric')."""
    mode = {"CONSTANT": "constant", "REFLECT": "reflect", "SYMMETRIC": "symmetric"}[mode.upper()]
    widths = [tuple(int(v) for v in p) for p in _np.asarray(paddings)]
    return _np.pad(_np.asarray(tensor), widths, mode=mode)


def meshgrid(*args, indexing="xy"):
    return _np.meshgrid(*args, indexing=indexing)


def concat(values, axis):
    return _np.concatenate([_np.asarray(v) for v in values], axis=axis)


def _conv_nd(nd, inputs, filters, padding):
    """channels-last input [N, *spatial, Cin], filter [*k, Cin, Cout], stride 1 -> torch conv."""
    import torch
    import torch.nn.functional as F
    if padding != "VALID":
        raise NotImplementedError("stand-in implements VALID only (all the reference uses)")
    x = _np.asarray(inputs)
    w = _np.asarray(filters).astype(x.dtype)
    perm_in = (0, nd + 1) + tuple(_np.arange(1, nd + 1))          # -> [N, Cin, *spatial]
    perm_w = (nd + 1, nd) + tuple(_np.arange(0, nd))              # -> [Cout, Cin, *k]
    xt = torch.from_numpy(_np.ascontiguousarray(_np.transpose(x, perm_in)))
    wt = torch.from_numpy(_np.ascontiguousarray(_np.transpose(w, perm_w)))
    out = (F.conv1d, F.conv2d, F.conv3d)[nd - 1](xt, wt)
    perm_out = (0,) + tuple(_np.arange(2, nd + 2)) + (1,)         # -> [N, *spatial, Cout]
    return _np.ascontiguousarray(out.permute(*[int(p) for p in perm_out]).numpy())


class _NN:
    @staticmethod
    def conv1d(input, filters, stride=1, padding="VALID"):       # noqa: A002
        assert stride in (1, [1], [1, 1, 1])
        return _conv_nd(1, input, filters, padding)

    @staticmethod
    def conv2d(input, filters, strides=1, padding="VALID"):      # noqa: A002
        assert strides in (1, [1, 1], [1, 1, 1, 1])
        return _conv_nd(2, input, filters, padding)

    @staticmethod
    def conv3d(input, filters, strides=1, padding="VALID"):      # noqa: A002
        assert strides in (1, [1, 1, 1], [1, 1, 1, 1, 1])
        return _conv_nd(3, input, filters, padding)


nn = _NN()
nest = object()          # data_utils.py:26,270 only binds the name


class _NameScope:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        return self.name

    def __exit__(self, *exc):
        return False


def name_scope(name):
    return _NameScope(name)


def cond(pred, true_fn, false_fn, name=None):
    return true_fn() if bool(pred) else false_fn()


def _pow(x, y):
    return _np.power(x, y)


_Math.pow = staticmethod(_pow)


class _Schedules:
    class LearningRateSchedule:
        def __init__(self):
            pass

    class ExponentialDecay(LearningRateSchedule):
        """Keras' documented rule: lr * rate ** (step / decay_steps), the exponent floored when
        staircase=True (kernel_train.py:57-58 builds it that way).  A stand-in for Keras, NOT
        reference code: only what WarmUp does with it is pinned."""

        def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False, name=None):
            self.initial_learning_rate = initial_learning_rate
            self.decay_steps = decay_steps
            self.decay_rate = decay_rate
            self.staircase = staircase

        def __call__(self, step):
            p = _np.asarray(step, dtype=float32) / float32(self.decay_steps)
            if self.staircase:
                p = _np.floor(p)
            return float32(self.initial_learning_rate) * _np.power(float32(self.decay_rate), p)


class _Optimizers:
    schedules = _Schedules


class _Keras:
    optimizers = _Optimizers


keras = _Keras


class _V1Train:
    @staticmethod
    def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
        return _Schedules.ExponentialDecay(learning_rate, decay_steps, decay_rate, staircase)(global_step)


class _V1:
    train = _V1Train


class _Compat:
    v1 = _V1


compat = _Compat
