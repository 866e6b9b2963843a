"""MI355X-native DINO-X model classes behind the reference's ``zoo.arch`` surface.

Same class names, constructor signatures, attribute names, sub-module names/types and
``state_dict`` keys as the reference (``zoo/arch.py`` of timlawrenz/DINO-X), so checkpoints, the hub
loader, ``encode()``, LoRA injection (``qkv/proj/fc1/fc2`` stay ``nn.Linear`` instances) and the
training loop work unchanged -- but every tensor op of ``forward``/``backward`` is a hand-written HIP
kernel from ``libdinox_hip.so`` (``dinox.ops``).  The modules only *hold* parameters; fused kernels
read the weights from them.  There is no CPU path: calling ``forward`` on CPU
tensors raises.

Reference lines mirrored:
  Attention           zoo/arch.py:28-54      Mlp                 zoo/arch.py:62-76
  TransformerBlock    zoo/arch.py:84-97      ScaleEmbedding      zoo/arch.py:105-140
  PatchViT            zoo/arch.py:148-238    DinoStudentTeacher  zoo/arch.py:246-261
  migrate_state_dict / needs_migration       zoo/arch.py:269-336
"""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from dinox import ops

__all__ = ["Attention", "Mlp", "TransformerBlock", "ScaleEmbedding", "PatchViT", "DinoStudentTeacher",
           "migrate_state_dict", "needs_migration"]


# ------------------------------------------------------------------------------------------
# leaf modules: parameter containers whose forward is a HIP kernel
# ------------------------------------------------------------------------------------------
class Linear(nn.Linear):
    """``nn.Linear`` whose product runs on the MFMA GEMM (``dinox_gemm``)."""

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, out_dtype=None) -> torch.Tensor:
        return ops.LinearFn.apply(x, self.weight, self.bias, residual, out_dtype)


class LayerNorm(nn.LayerNorm):
    """``nn.LayerNorm`` on the fp32 residual stream (``dinox_layernorm_*``).  ``out_dtype=None`` writes the
    GEMM operand dtype of the current mode (bf16 under autocast), ``torch.float32`` keeps fp32."""

    def forward(self, x: torch.Tensor, out_dtype=None) -> torch.Tensor:
        if x.dtype != torch.float32:
            x = x.float()
        return ops.LayerNormFn.apply(x, self.weight, self.bias, out_dtype or ops.current_dtype(), self.eps)


class GELU(nn.GELU):
    """Exact-erf GELU.  Only reached when fc1/fc2 were wrapped (e.g. by LoRA) and the fused MLP path
    cannot be used; the fused path applies GELU inside the fc1 GEMM epilogue."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.GeluFn.apply(x)


class Attention(nn.Module):
    """Multi-head self-attention; ``qkv``/``proj`` are ``nn.Linear`` instances for peft targeting."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = True) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = Linear(dim, dim)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        qkv = self.qkv(x)                                     # packed [B, N, 3, h, d]; never permuted or copied
        o = ops.AttentionCoreFn.apply(qkv, self.num_heads)    # [B, N, h*d]
        if type(self.proj) is Linear:
            return self.proj(o, residual=residual)
        y = self.proj(o)
        return y if residual is None else residual + y


class Mlp(nn.Module):
    """fc2(GELU(fc1(x))); ``fc1``/``fc2`` are ``nn.Linear`` instances for peft targeting."""

    def __init__(self, dim: int, mlp_ratio: float = 4.0) -> None:
        super().__init__()
        hidden = int(dim * mlp_ratio)
        self.fc1 = Linear(dim, hidden)
        self.act = GELU()
        self.fc2 = Linear(hidden, dim)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        if type(self.fc1) is Linear and type(self.fc2) is Linear:
            return ops.MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual, None)
        y = self.fc2(self.act(self.fc1(x)))
        return y if residual is None else residual + y


class TransformerBlock(nn.Module):
    """Pre-norm block: x += attn(norm1(x)); x += mlp(norm2(x)).  The residual adds are fused into the
    proj / fc2 GEMM epilogues, the LayerNorm outputs are written directly in the GEMM operand dtype."""

    def __init__(self, dim: int, heads: int, mlp_ratio: float = 4.0) -> None:
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn = Attention(dim, heads)
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, mlp_ratio)

    def _fusable(self) -> bool:
        a, m = self.attn, self.mlp
        return (type(self.norm1) is LayerNorm and type(self.norm2) is LayerNorm and type(a) is Attention and type(m) is Mlp
                and type(a.qkv) is Linear and type(a.proj) is Linear and type(m.fc1) is Linear and type(m.fc2) is Linear
                and self.norm1.eps == self.norm2.eps)

    def forward_chained(self, x: torch.Tensor, pre_ln, next_norm: "LayerNorm", next_dtype=None):
        """The block as one autograd node that takes norm1(x) from the producer of x (``pre_ln`` = (y, mean, rstd) or None) and
        also returns ``next_norm`` applied to its output -- in bf16 mode at width 384 both LayerNorms of the chain come out of the
        proj / fc2 products' epilogues (ops.linear_residual_ln) instead of separate launches.  -> (x_out, (y, mean, rstd))."""
        a, m = self.attn, self.mlp
        nl = (next_norm.weight.detach(), next_norm.bias.detach(), next_norm.eps, next_dtype)
        out = ops.block_fn(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                a.num_heads, self.norm1.eps, pre_ln, nl)
        return out[0], (out[1], out[2], out[3])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._fusable():      # the whole block as ONE autograd node (ops.BlockFn); weights are read from the sub-modules
            a, m = self.attn, self.mlp
            return ops.block_fn(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                     self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                     a.num_heads, self.norm1.eps)
        # a sub-module was replaced (e.g. LoRA-wrapped): compose the per-op nodes instead
        x = self.attn(self.norm1(x), residual=x)
        x = self.mlp(self.norm2(x), residual=x)
        return x


class ScaleEmbedding(nn.Module):
    """(pixel_spacing_x, pixel_spacing_y, slice_thickness) in mm -> (B, 1, embed_dim).

    ``mlp`` keeps the reference layout ``Sequential(Linear(3,h), GELU, Linear(h,D), LayerNorm(D))`` with
    a zero-initialised output projection, so a fresh module is a no-op and old checkpoints resume
    identically; the forward is one fused HIP kernel over those parameters."""

    def __init__(self, embed_dim: int) -> None:
        super().__init__()
        hidden = max(embed_dim // 4, 16)
        self.mlp = nn.Sequential(
            nn.Linear(3, hidden),
            nn.GELU(),
            nn.Linear(hidden, embed_dim),
            nn.LayerNorm(embed_dim),
        )
        nn.init.zeros_(self.mlp[2].weight)
        nn.init.zeros_(self.mlp[2].bias)

    def forward(self, spacing: torch.Tensor) -> torch.Tensor:
        m = self.mlp
        return ops.ScaleEmbedFn.apply(spacing, m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[3].weight, m[3].bias, m[3].eps)


# ------------------------------------------------------------------------------------------
# PatchViT
# ------------------------------------------------------------------------------------------
class PatchViT(nn.Module):
    """Patch ViT with CLS token, register tokens and optional ScaleEmbedding.

    Token order is [CLS, patches..., registers...]; ``forward`` returns all tokens after the final
    LayerNorm in fp32, shape (B, 1 + P + num_registers, dim)."""

    def __init__(
        self,
        img_size: int = 224,
        patch: int = 16,
        dim: int = 384,
        depth: int = 6,
        heads: int = 6,
        mlp_ratio: float = 4.0,
        use_grad_checkpoint: bool = False,
        num_registers: int = 4,
        scale_aware: bool = False,
    ) -> None:
        super().__init__()
        assert img_size % patch == 0
        self.img_size, self.patch, self.dim = img_size, patch, dim
        self.use_grad_checkpoint = use_grad_checkpoint
        self.num_registers = num_registers
        self.scale_aware = scale_aware

        # Parameter container only: the convolution itself runs as unfold + MFMA GEMM (ops.TokensFn).
        self.patch_embed = nn.Conv2d(3, dim, kernel_size=patch, stride=patch, bias=True)
        n_patches = (img_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + n_patches, dim))
        if num_registers > 0:
            self.registers = nn.Parameter(torch.zeros(1, num_registers, dim))
        if scale_aware:
            self.scale_embed = ScaleEmbedding(dim)
        self.blocks = nn.ModuleList([TransformerBlock(dim, heads, mlp_ratio) for _ in range(depth)])
        self.norm = LayerNorm(dim)

        # Same visiting order and draws as the reference (its initialiser re-draws the token tensors once
        # per visited sub-module), so a given torch seed yields bit-identical initial weights.
        self.apply(self._init_weights)
        if scale_aware:
            nn.init.zeros_(self.scale_embed.mlp[2].weight)
            nn.init.zeros_(self.scale_embed.mlp[2].bias)

    def _init_weights(self, m: nn.Module) -> None:
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        nn.init.trunc_normal_(self.pos_embed, std=0.1)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.trunc_normal_(self.patch_embed.weight, std=0.02)
        if self.num_registers > 0:
            nn.init.trunc_normal_(self.registers, std=0.02)

    def forward(self, x: torch.Tensor, spacing: Optional[torch.Tensor] = None) -> torch.Tensor:
        g = x.shape[-1] // self.patch
        if x.shape[-2] != x.shape[-1] or x.shape[-1] % self.patch:
            raise ValueError(f"input {tuple(x.shape[-2:])} is not a square multiple of the {self.patch}-pixel patch")
        if x.shape[0] == 0:
            ops._need_cuda(x, self.pos_embed)
            # An empty batch launches nothing (the kernels take no null operands).  Like the reference's modules it returns
            # (0, 1 + P + R, D) in fp32, attached to the parameters so that a backward pass leaves zero gradients, not None.
            anchor = sum((q.sum() * 0 for q in self.parameters() if q.requires_grad), torch.zeros((), device=x.device))
            return anchor.expand(0, 1 + g * g + self.num_registers, self.dim) * 1.0
        scale = None
        if self.scale_aware and spacing is not None:
            scale = self.scale_embed(spacing)
        regs = self.registers if self.num_registers > 0 else None
        pos = self.pos_embed
        if g * g != pos.shape[1] - 1:       # extension (multi-crop local views): the reference has one input size only
            pos = ops.interp_pos(pos, g)
        t = ops.TokensFn.apply(x, self.patch_embed.weight, self.patch_embed.bias, self.cls_token, pos, regs, scale, self.patch)
        chain = (not (self.use_grad_checkpoint and self.training) and type(self.norm) is LayerNorm and len(self.blocks) > 0
                 and all(type(b) is TransformerBlock and b._fusable() for b in self.blocks)
                 and all(b.norm1.eps == self.norm.eps for b in self.blocks))
        if chain:
            # every LayerNorm but the first rides in the epilogue of the product that wrote its input: block i hands
            # norm1_{i+1}(x) to block i + 1, the last block hands over the final norm (fp32: the features)
            pre = None
            for i, blk in enumerate(self.blocks):
                last = i + 1 == len(self.blocks)
                t, pre = blk.forward_chained(t, pre, self.norm if last else self.blocks[i + 1].norm1, torch.float32 if last else None)
            return ops.LayerNormPrecomputedFn.apply(t, self.norm.weight, self.norm.bias, *pre)
        for blk in self.blocks:
            if self.use_grad_checkpoint and self.training:
                t = torch.utils.checkpoint.checkpoint(blk, t, use_reentrant=False, context_fn=ops.checkpoint_contexts)
            else:
                t = blk(t)
        return self.norm(t, out_dtype=torch.float32)


class DinoHead(nn.Sequential):
    """Linear(D,D) -> GELU -> Linear(D,out) as ONE fused call (keys ``0.*`` / ``2.*`` as in the reference)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        l0, l2 = self[0], self[2]
        return ops.MlpFn.apply(x, l0.weight, l0.bias, l2.weight, l2.bias, None, torch.float32)


class DinoStudentTeacher(nn.Module):
    """Backbone + DINO projection head; the training loop calls ``.backbone`` and ``.head`` separately."""

    def __init__(self, backbone: nn.Module, out_dim: int = 8192) -> None:
        super().__init__()
        self.backbone = backbone
        self.head = DinoHead(
            Linear(backbone.dim, backbone.dim),
            GELU(),
            Linear(backbone.dim, out_dim),
        )

    def forward(self, x: torch.Tensor, spacing: Optional[torch.Tensor] = None) -> torch.Tensor:
        feats = self.backbone(x, spacing=spacing)
        return self.head(feats[:, 0])


# ------------------------------------------------------------------------------------------
# state-dict migration (old nn.MultiheadAttention / nn.Sequential-MLP checkpoints)
# ------------------------------------------------------------------------------------------
_RENAMES = (
    # attention: any prefix ending in ".attn"
    (re.compile(r"^(?P<pre>.+\.attn)\.in_proj_(?P<kind>weight|bias)$"), r"\g<pre>.qkv.\g<kind>"),
    (re.compile(r"^(?P<pre>.+\.attn)\.out_proj\.(?P<kind>weight|bias)$"), r"\g<pre>.proj.\g<kind>"),
    # transformer-block MLP only (scale_embed.mlp is a real nn.Sequential and keeps numeric keys)
    (re.compile(r"^(?P<pre>(?:.*\.)?blocks\.\d+\.mlp)\.0\.(?P<kind>weight|bias)$"), r"\g<pre>.fc1.\g<kind>"),
    (re.compile(r"^(?P<pre>(?:.*\.)?blocks\.\d+\.mlp)\.2\.(?P<kind>weight|bias)$"), r"\g<pre>.fc2.\g<kind>"),
)


def _new_key(key: str) -> str:
    for pat, repl in _RENAMES:
        if pat.match(key):
            return pat.sub(repl, key)
    return key


def migrate_state_dict(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Return a new dict with old-format keys renamed to the timm-style ones; other keys pass through."""
    return OrderedDict((_new_key(k), v) for k, v in state_dict.items())


def needs_migration(state_dict: Dict[str, torch.Tensor]) -> bool:
    return any(_new_key(k) != k for k in state_dict)
