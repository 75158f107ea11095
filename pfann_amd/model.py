"""Host mirror of the reference's encoder interface (model.py:132-153):
`FpNetwork(d, h, u, F, T, params)`, `load_state_dict`, `forward(x, norm=True)`.
The arithmetic runs in hand-written HIP kernels (pfann_amd/csrc/encoder.hip)."""
import torch

from .engine import Engine


class FpNetwork:
    def __init__(self, d, h, u, F, T, params, device=0, engine=None, max_batch=512):
        """`params` is the config's "model" dict as in the reference.  Either pass a shared
        `engine` (built from the full config) or let this build a private one."""
        if engine is None:
            # the reference's constructor (model.py:132-146) sees no front-end settings: an ENCODER-ONLY context.  Its
            # front-end fields only have to be self-consistent (any length with ceil(len/hop) = 1 + len//hop = T, builder.py:50-51);
            # melspec / embed_* on such a context raise.
            full = dict(sample_rate=8000, segment_size=((T - 1) * 256 + 128) / 8000.0, stft_n=1024, stft_hop=256, n_mels=F,
                        f_min=300, f_max=4000, model=dict(params, d=d, h=h, u=u))
            engine = Engine(full, device, max_batch, encoder_only=True)
        assert (engine.d, engine.h, engine.u, engine.F, engine.T) == (d, h, u, F, T)
        self.engine = engine
        self.d, self.h, self.u = d, h, u

    # torch.nn.Module-shaped conveniences used by the reference's scripts
    def to(self, device):
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def load_state_dict(self, state_dict, strict=True):
        self.engine.load_state_dict(state_dict)

    def forward(self, x, norm=True):
        return self.engine.encode(x, norm=norm)

    def __call__(self, x, norm=True):
        return self.forward(x, norm=norm)


def load_model(params, model_dir_or_file, device=0, engine=None, max_batch=512):
    """FpNetwork with weights from <dir>/model.pt (builder.py:55-56, matcher.py:60-61)."""
    import os
    from .synth import model_dims
    path = model_dir_or_file
    if os.path.isdir(path):
        path = os.path.join(path, "model.pt")
    d, h, u, F, T = model_dims(params)
    if engine is None:
        engine = Engine(params, device, max_batch)
    net = FpNetwork(d, h, u, F, T, params["model"], engine=engine)
    net.load_state_dict(torch.load(path, map_location="cpu"))
    return net
