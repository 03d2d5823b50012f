"""Host-side holders of the G / H (v2: g_net / e_net) dense networks.

They carry the weights (numpy, [in, out] like TensorFlow) and the layer structure the reference's
network classes define; the arithmetic runs in librt_is.so (rt_mlp_forward), never on the host.

Reference: model.Generator (/root/reference/model.py:53-92) and BaseFullyConnectedNet
(/root/reference/src/pyroundtrip/model.py:4-50).
"""
import numpy as np


class DenseNet(object):
    """nb_layers x (dense + leaky-ReLU 0.2) followed by a linear dense layer."""

    def __init__(self, input_dim, output_dim, name, nb_units, alpha=0.2, rng=None):
        nb_units = [int(u) for u in nb_units]
        if len(nb_units) < 1:
            # src/pyroundtrip/model.py:37-47 also needs at least one hidden layer
            raise ValueError("at least one hidden layer is required")
        self.input_dim = int(input_dim)
        self.output_dim = int(output_dim)
        self.name = name
        self.model_name = name
        self.nb_units = nb_units
        self.alpha = float(alpha)
        self.version = 0          # bumped whenever the weights change (the engine re-uploads lazily)
        self._forward = None      # bound by the model object: callable(float32 (N,in)) -> float32 (N,out)
        rng = rng if rng is not None else np.random.RandomState()
        dims = [self.input_dim] + nb_units + [self.output_dim]
        self.weights, self.biases = [], []
        for fan_in, fan_out in zip(dims[:-1], dims[1:]):
            # Glorot uniform: the default kernel initializer of tcl.fully_connected and keras Dense
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            self.weights.append(rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32))
            self.biases.append(np.zeros(fan_out, dtype=np.float32))

    @property
    def dims(self):
        return [self.input_dim] + self.nb_units + [self.output_dim]

    def set_weights(self, weights, biases):
        """weights[l]: (dims[l], dims[l+1]) ; biases[l]: (dims[l+1],)"""
        dims = self.dims
        if len(weights) != len(dims) - 1 or len(biases) != len(dims) - 1:
            raise ValueError("%s expects %d layers, got %d weights / %d biases"
                             % (self.name, len(dims) - 1, len(weights), len(biases)))
        ws, bs = [], []
        for l, (w, b) in enumerate(zip(weights, biases)):
            w = np.ascontiguousarray(w, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
            if w.shape != (dims[l], dims[l + 1]) or b.shape != (dims[l + 1],):
                raise ValueError("%s layer %d: expected kernel %r and bias %r, got %r and %r"
                                 % (self.name, l, (dims[l], dims[l + 1]), (dims[l + 1],), w.shape, b.shape))
            ws.append(w)
            bs.append(b)
        self.weights, self.biases = ws, bs
        self.version += 1

    def get_weights(self):
        out = []
        for w, b in zip(self.weights, self.biases):
            out += [w, b]
        return out

    def __call__(self, x):
        if self._forward is None:
            raise RuntimeError("%s is not attached to a Roundtrip model (no device engine)" % self.name)
        x = np.asarray(x)
        assert x.shape[-1] == self.input_dim
        return self._forward(np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.input_dim))


class Generator(DenseNet):
    """v1 spelling: model.Generator(input_dim, output_dim, name, nb_layers=2, nb_units=256)
    (/root/reference/model.py:54-59): nb_layers hidden layers of nb_units each."""

    def __init__(self, input_dim, output_dim, name, nb_layers=2, nb_units=256, rng=None):
        super(Generator, self).__init__(input_dim, output_dim, name, [nb_units] * int(nb_layers), rng=rng)
        self.nb_layers = int(nb_layers)


class BaseFullyConnectedNet(DenseNet):
    """v2 spelling: BaseFullyConnectedNet(input_dim, output_dim, model_name, nb_units=[...], batchnorm=False)
    (/root/reference/src/pyroundtrip/model.py:7)."""

    def __init__(self, input_dim, output_dim, model_name, nb_units=(256, 256, 256), batchnorm=False, rng=None):
        if batchnorm:
            raise NotImplementedError("batchnorm=True is outside the importance-sampling path (off by default "
                                      "in the reference, src/pyroundtrip/model.py:7)")
        super(BaseFullyConnectedNet, self).__init__(input_dim, output_dim, model_name, list(nb_units), rng=rng)
        self.batchnorm = False
