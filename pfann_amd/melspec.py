"""Host mirror of the reference's mel front-end interface (datautil/melspec.py):
`build_mel_spec_layer(params)` returns a callable `mel(x[..., 8000]) -> [..., 256, 32]`
that runs the fused HIP STFT/mel/log kernel."""
from .engine import Engine, mel_filterbank  # noqa: F401


class MelSpec:
    """MelSpec.forward (melspec.py:33-50) on the MI355X."""

    def __init__(self, params, device=0, engine=None):
        self.engine = engine if engine is not None else Engine(params, device)

    def to(self, device):
        return self

    def forward(self, x):
        return self.engine.melspec(x)

    __call__ = forward


def build_mel_spec_layer(params, device=0, engine=None):
    """Same name/arguments as melspec.py:52-63 (plus optional device / shared engine)."""
    return MelSpec(params, device, engine)
