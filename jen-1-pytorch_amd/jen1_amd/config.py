"""Model / diffusion configuration values and the derived parameter schema.

The values restate /root/reference/utils/config.py:23-33 (GDM_Config),
:50-74 (ModelConfig) and :76-82 (OptimizerConfig).  ``UNetSpec`` turns the
``UNetCFG1d`` constructor kwargs (/root/reference/jen1/model/model.py:14-37,
271-277) into the level table of SURVEY.md Appendix B and the ``state_dict``
key schema of Appendix C, so weights can be created, packed and checked
without importing the reference.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple


def full_model_config() -> dict:
    """kwargs of the full JEN-1 UNetCFG1d (reference utils/config.py:50-74)."""
    return dict(
        in_channels=128, channels=128,
        multipliers=[1, 1, 1, 2, 2, 4, 4, 4, 8, 8],
        factors=[1, 4, 4, 4, 2, 2, 2, 2, 2],
        num_blocks=[1, 3, 3, 3, 3, 3, 3, 3, 1],
        attentions=[0, 0, 0, 1, 1, 1, 1, 1, 1],
        patch_size=1, resnet_groups=8, use_context_time=True,
        kernel_multiplier_downsample=2, use_nearest_upsample=False,
        use_skip_scale=True, use_snake=False, use_stft=False, use_stft_context=False,
        use_xattn_time=True, out_channels=128, context_features=None,
        context_features_multiplier=4, context_channels=[129],
        context_embedding_features=1024, context_embedding_max_length=128,
        attention_heads=8, attention_multiplier=1,
    )


def tiny_model_config() -> dict:
    """BASELINE.json configs[0]: tiny 1D-UNet, d=64, 2 res blocks."""
    return dict(
        in_channels=128, out_channels=128, channels=64,
        multipliers=[1, 1, 2], factors=[1, 2], num_blocks=[2, 2], attentions=[0, 1, 1],
        patch_size=1, resnet_groups=8, context_channels=[129],
        attention_heads=8, attention_multiplier=1, use_xattn_time=True,
        context_embedding_features=1024, context_embedding_max_length=128,
    )


def linear_param_shapes(name: str, o: int, i: int, bias: bool = True):
    out = [(name + ".weight", (o, i))]
    if bias:
        out.append((name + ".bias", (o,)))
    return out


def res_param_shapes(n: str, c_in: int, c_out: int, mapping_features: int):
    """ResnetBlock1d parameters (reference blocks.py:168-217)."""
    S = [(f"{n}.block1.groupnorm.weight", (c_in,)), (f"{n}.block1.groupnorm.bias", (c_in,)),
         (f"{n}.block1.project.conv.weight", (c_out, c_in, 3)), (f"{n}.block1.project.conv.bias", (c_out,))]
    S += linear_param_shapes(f"{n}.to_scale_shift.to_scale_shift.1", 2 * c_out, mapping_features)
    S += [(f"{n}.block2.groupnorm.weight", (c_out,)), (f"{n}.block2.groupnorm.bias", (c_out,)),
          (f"{n}.block2.project.conv.weight", (c_out, c_out, 3)), (f"{n}.block2.project.conv.bias", (c_out,))]
    if c_in != c_out:
        S += [(f"{n}.to_out.conv.weight", (c_out, c_in, 1)), (f"{n}.to_out.conv.bias", (c_out,))]
    return S


def attention_param_shapes(n: str, c: int, heads: int, head_features: int, context_features: int):
    """Attention parameters (reference blocks.py:383-413)."""
    mid = heads * head_features
    S = [(f"{n}.norm.weight", (c,)), (f"{n}.norm.bias", (c,)),
         (f"{n}.norm_context.weight", (context_features,)), (f"{n}.norm_context.bias", (context_features,))]
    S += linear_param_shapes(f"{n}.to_q", mid, c, bias=False)
    S += linear_param_shapes(f"{n}.to_kv", 2 * mid, context_features, bias=False)
    S += linear_param_shapes(f"{n}.attention.to_out", c, mid)
    return S


def transformer_param_shapes(n: str, c: int, heads: int, head_features: int, multiplier: int,
                             context_features: int, num_layers: int = 1):
    """Transformer1d parameters (reference blocks.py:497-526, 454-481)."""
    S = [(f"{n}.group_norm.weight", (c,)), (f"{n}.group_norm.bias", (c,)),
         (f"{n}.conv1d.conv.weight", (c, c, 1)), (f"{n}.conv1d.conv.bias", (c,))]
    for l in range(num_layers):
        b = f"{n}.blocks.{l}"
        S += attention_param_shapes(f"{b}.attention", c, heads, head_features, c)
        S += attention_param_shapes(f"{b}.cross_attention", c, heads, head_features, context_features)
        S += linear_param_shapes(f"{b}.feed_forward.0", c * multiplier, c)
        S += linear_param_shapes(f"{b}.feed_forward.2", c, c * multiplier)
    return S


@dataclass
class GDMConfig:
    """reference utils/config.py:23-33."""
    steps: int = 1000
    noise_schedule: str = "linear"
    objective: str = "noise"
    loss_type: str = "l2"
    cfg_dropout_proba: float = 0.2
    embedding_scale: float = 0.8
    batch_cfg: bool = True
    scale_cfg: bool = True


@dataclass
class VDMConfig:
    """reference utils/config.py:35-42."""
    loss_type: str = "l2"
    cfg_dropout_proba: float = 0.2
    embedding_scale: float = 0.8
    batch_cfg: bool = True
    scale_cfg: bool = True


@dataclass
class OptimizerConfig:
    """reference utils/config.py:76-82, :96."""
    lr: float = 3e-5
    beta_1: float = 0.9
    beta_2: float = 0.95
    weight_decay: float = 0.1
    grad_clip: float = 0.7
    grad_accum_every: int = 10


@dataclass
class TrainConfig:
    """the fields of reference utils/config.py:84-100 (``Config``) that the trainer reads."""
    save_dir: str = ""
    use_fp16: bool = False
    tasks: List[str] = field(default_factory=lambda: ["text_guided", "music_inpaint", "music_cont"])
    num_epoch: int = 100
    eval_interval: int = 30
    grad_accum_every: int = 10
    device: str = "cuda"
    diffusion_type: str = "gdm"
    optimizer_config: OptimizerConfig = field(default_factory=OptimizerConfig)


@dataclass
class ResSpec:
    name: str
    c_in: int
    c_out: int
    groups: int

    @property
    def has_shortcut(self) -> bool:
        return self.c_in != self.c_out


@dataclass
class TransformerSpec:
    name: str
    channels: int
    heads: int
    head_features: int
    multiplier: int
    context_features: int
    num_layers: int = 1


@dataclass
class DownSpec:
    name: str
    c_in: int
    c_out: int
    factor: int
    kernel: int
    blocks: List[ResSpec]
    transformer: Optional[TransformerSpec]


@dataclass
class UpSpec:
    name: str
    c_in: int           # channels of the blocks (deep side)
    c_out: int          # channels after the upsample
    factor: int
    blocks: List[ResSpec]
    transformer: Optional[TransformerSpec]


class UNetSpec:
    """Static structure of a UNetCFG1d instance (no tensors)."""

    _KNOWN = {
        "in_channels", "channels", "multipliers", "factors", "num_blocks", "attentions",
        "patch_size", "resnet_groups", "use_context_time", "kernel_multiplier_downsample",
        "use_nearest_upsample", "use_skip_scale", "use_snake", "use_stft", "use_stft_context",
        "out_channels", "context_features", "context_features_multiplier", "context_channels",
        "context_embedding_features", "context_embedding_max_length", "use_xattn_time",
        "attention_heads", "attention_multiplier", "attention_features",
    }

    def __init__(self, **kw):
        unknown = set(kw) - self._KNOWN
        # reference model.py:110 asserts on unknown kwargs
        assert not unknown, f"Unknown arguments: {', '.join(sorted(unknown))}"
        self.kwargs = dict(kw)
        g = kw.get
        self.in_channels: int = g("in_channels")
        self.channels: int = g("channels")
        self.multipliers: Sequence[int] = list(g("multipliers"))
        self.factors: Sequence[int] = list(g("factors"))
        self.num_blocks: Sequence[int] = list(g("num_blocks"))
        self.attentions: Sequence[int] = list(g("attentions"))
        self.patch_size: int = g("patch_size", 1)
        self.resnet_groups: int = g("resnet_groups", 8)
        self.use_context_time: bool = g("use_context_time", True)
        self.kmul: int = g("kernel_multiplier_downsample", 2)
        self.use_skip_scale: bool = g("use_skip_scale", True)
        self.out_channels: int = g("out_channels") or self.in_channels
        self.context_features = g("context_features")
        self.cf_mult: int = g("context_features_multiplier", 4)
        self.context_channels: List[int] = list(g("context_channels") or [])
        self.ctx_features: int = g("context_embedding_features")
        self.ctx_max_length: int = g("context_embedding_max_length")
        self.use_xattn_time: bool = g("use_xattn_time", False)
        self.heads: int = g("attention_heads")
        self.att_mult: int = g("attention_multiplier")
        self.att_features = g("attention_features")

        # features this build does not carry (dead/disabled
        # in the reference config): fail loudly instead of silently diverging.
        for flag in ("use_snake", "use_stft", "use_stft_context", "use_nearest_upsample"):
            assert not g(flag, False), f"{flag}=True is outside the JEN-1 hot path (SURVEY.md section 2)"
        assert self.context_features is None, "global_cond / context_features is unused on the JEN-1 path"
        assert self.patch_size == 1, "patch_size != 1 is unused on the JEN-1 path"
        assert self.use_context_time, "use_context_time=False is unused on the JEN-1 path"
        assert self.kmul % 2 == 0, "Kernel multiplier must be even"   # blocks.py:58

        L = len(self.multipliers) - 1
        self.num_layers = L
        assert len(self.factors) == L and len(self.attentions) >= L and len(self.num_blocks) == L  # model.py:66-70
        cc = self.context_channels + [0] * (L + 1 - len(self.context_channels))
        assert all(c == 0 for c in cc[1:]), "context channels are only injected at layer 0 on this path"
        self.ctx_ch0 = cc[0]
        self.mapping_features = self.channels * self.cf_mult
        self.ctx_len = self.ctx_max_length + (1 if self.use_xattn_time else 0)

        C = [self.channels * m for m in self.multipliers]
        G = self.resnet_groups

        def tr(name, ch, n):
            if n <= 0:
                return None
            hf = self.att_features or ch // self.heads
            return TransformerSpec(name, ch, self.heads, hf, self.att_mult, self.ctx_features, n)

        self.to_in = ResSpec("to_in.block", self.in_channels + self.ctx_ch0, C[0], 1)
        self.to_out = ResSpec("to_out.block", C[0], self.out_channels, 1)
        self.downs: List[DownSpec] = []
        for i in range(L):
            blocks = [ResSpec(f"downsamples.{i}.blocks.{j}", C[i + 1], C[i + 1], G) for j in range(self.num_blocks[i])]
            self.downs.append(DownSpec(f"downsamples.{i}", C[i], C[i + 1], self.factors[i],
                                       self.factors[i] * self.kmul + 1, blocks,
                                       tr(f"downsamples.{i}.transformer", C[i + 1], self.attentions[i])))
        cb = C[-1]
        self.bott_pre = ResSpec("bottleneck.pre_block", cb, cb, G)
        self.bott_post = ResSpec("bottleneck.post_block", cb, cb, G)
        self.bott_tr = tr("bottleneck.transformer", cb, self.attentions[-1])      # model.py:147
        self.ups: List[UpSpec] = []
        for idx, i in enumerate(reversed(range(L))):
            nl = self.num_blocks[i] + (1 if self.attentions[i] else 0)       # model.py:159
            blocks = [ResSpec(f"upsamples.{idx}.blocks.{j}", 2 * C[i + 1], C[i + 1], G) for j in range(nl)]
            self.ups.append(UpSpec(f"upsamples.{idx}", C[i + 1], C[i], self.factors[i], blocks,
                                   tr(f"upsamples.{idx}.transformer", C[i + 1], self.attentions[i])))

    # ------------------------------------------------------------------ schema
    def res_blocks(self) -> List[ResSpec]:
        out = [self.to_in]
        for d in self.downs:
            out += d.blocks
        out += [self.bott_pre, self.bott_post]
        for u in self.ups:
            out += u.blocks
        out.append(self.to_out)
        return out

    def transformers(self) -> List[TransformerSpec]:
        out = [d.transformer for d in self.downs if d.transformer]
        if self.bott_tr:
            out.append(self.bott_tr)
        out += [u.transformer for u in self.ups if u.transformer]
        return out

    def param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(key, shape) in reference ``state_dict`` naming."""
        S: List[Tuple[str, Tuple[int, ...]]] = []
        mf = self.mapping_features
        half = self.channels // 2

        def lin(name, o, i, bias=True):
            S.extend(linear_param_shapes(name, o, i, bias))

        lin("to_mapping.0", mf, mf)
        lin("to_mapping.2", mf, mf)
        S.append(("to_time.0.0.weights", (half,)))
        lin("to_time.0.1", mf, self.channels + 1)

        def res(r: ResSpec):
            S.extend(res_param_shapes(r.name, r.c_in, r.c_out, mf))

        def trf(t: TransformerSpec):
            S.extend(transformer_param_shapes(t.name, t.channels, t.heads, t.head_features, t.multiplier,
                                              t.context_features, t.num_layers))

        res(self.to_in)
        for d in self.downs:
            S.append((f"{d.name}.downsample.conv.weight", (d.c_out, d.c_in, d.kernel)))
            S.append((f"{d.name}.downsample.conv.bias", (d.c_out,)))
            for r in d.blocks:
                res(r)
            if d.transformer:
                trf(d.transformer)
        res(self.bott_pre)
        if self.bott_tr:
            trf(self.bott_tr)
        res(self.bott_post)
        for u in self.ups:
            for r in u.blocks:
                res(r)
            if u.transformer:
                trf(u.transformer)
            if u.factor == 1:
                S.append((f"{u.name}.upsample.weight", (u.c_out, u.c_in, 3)))
            else:
                S.append((f"{u.name}.upsample.weight", (u.c_in, u.c_out, 2 * u.factor)))
            S.append((f"{u.name}.upsample.bias", (u.c_out,)))
        res(self.to_out)
        if self.use_xattn_time:
            S.append(("to_time_embedding.0.0.weights", (half,)))
            lin("to_time_embedding.0.1", self.ctx_features, self.channels + 1)
        S.append(("fixed_embedding.embedding.weight", (self.ctx_len, self.ctx_features)))
        return S

    def num_params(self) -> int:
        n = 0
        for _, s in self.param_shapes():
            p = 1
            for d in s:
                p *= d
            n += p
        return n

    # ------------------------------------------------------------------ lengths
    def level_lengths(self, T: int) -> List[int]:
        """Sequence length after each down level: ceil(L/f) (blocks.py:34-53, stride f)."""
        out = [T]
        for f in self.factors:
            out.append((out[-1] + f - 1) // f)
        return out
