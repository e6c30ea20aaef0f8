"""SelfAttentionBlock (ganslate/nn/attention.py:12-47) — the SAGAN attention layer on N = D*W*H voxels: query / key = 1x1x1
convs to C/8 channels, value = 1x1x1 conv to C channels, attention = softmax_j(q_i . k_j), out = gamma * (attention @ v) + x
with gamma initialised to zero.

In this package the block is not a module of its own: a network executor applies it to an NDHWC bf16 activation through
`ops.attn_forward / ops.attn_backward` (csrc/attn.hip: batched MFMA GEMMs + a row softmax), and its seven parameters live in
the executor's flat fp32 master buffer as `Extra`s under the reference's state-dict names (`<prefix>.gamma`,
`<prefix>.query_conv.weight`, ...). `NativeNet` applies it to the output of a node marked `Node(attn=<prefix>)`
(SelfAttentionPatchGAN3D), `SelfAttentionVnet3D` to the outputs of its down blocks."""
from .native.net import ATTN_KEYS, attention_extras  # noqa: F401
