"""CPU restatement of the reference's channel-level perturbation (TEST INFRASTRUCTURE ONLY).

Follows code/networks/FilterDropout.py of the reference:
  perform_dropout      :45-89    scores_dropoutV2  :116-138    drop_based_on_prob :140-160
and the `dropout=True` branch of DualDecoder.forward (code/networks/unet.py:277-292).

The reference draws its masks with torch.bernoulli / Binomial.sample / nn.Dropout2d / random.randint; here every
draw is an explicit input -- bernoulli(q) == (u < q) on a caller-supplied uniform tensor, `branch` == the
random.randint(0, 1) of drop_based_on_prob -- so that the HIP path and this file can be driven with identical
randomness.  Pinned against the imported reference (its torch.bernoulli replaced by the same u < q rule while the
vectors are generated) by tests/golden/filter_dropout.npz, see oracle/gen_golden.py.
"""
import math

import torch

from . import nets


def fd_inputs(B=4):
    """Seeded inputs of the FilterDropout vectors (regenerated identically by the tests)."""
    g = torch.Generator().manual_seed(900)
    C = (16, 32, 64, 128, 256)
    sp = (8, 4, 4, 2, 2)
    feats = [torch.rand(B, c, s, s, generator=g) + 0.05 for c, s in zip(C, sp)]
    scores = [torch.randn(c, generator=g) for c in C]
    scores[2] = torch.zeros(C[2])                                   # an all-zero level -> Dropout2d fallback
    uniforms = [(torch.rand(B // 2, c, generator=g), torch.rand(B // 2, c, generator=g)) for c in C]
    return feats, scores, uniforms


def drop_based_on_prob(drop_probs, comp, u1, u2, branch=0):
    """FilterDropout.py:140-160.  drop_probs, u1, u2: [U, C].  Returns the two rescaled masks [U, C, 1, 1]."""
    keep = 1 - drop_probs
    if comp:
        q1, q2 = (keep, drop_probs) if branch == 0 else (drop_probs, keep)
    else:
        q1, q2 = keep, keep
    m1 = (u1 < q1).float()[..., None, None]
    m2 = (u2 < q2).float()[..., None, None]
    return m1 * m1.numel() / m1.sum(), m2 * m2.numel() / m2.sum()


def drop_probs(grad_sim, activation, kind="sigmoid"):
    """The probability part of scores_dropoutV2 (FilterDropout.py:121-134)."""
    scores = grad_sim.unsqueeze(0).expand(activation.size(0), activation.size(1)) * activation
    sigma = torch.std(scores, dim=1, keepdim=True)
    mean = torch.mean(scores, dim=1, keepdim=True)
    if kind == "gauss":
        z = (scores - mean) / (sigma * 2.0 + 1e-8)
        return torch.clamp(0.5 * (1 + torch.erf(z / math.sqrt(2))), 0.0, 1.0)
    z = (scores - mean) / (sigma + 1e-8)
    return torch.sigmoid(-z * 2.0)


def scores_dropout_v2(grad_sim, activation, comp, kind, u1, u2, branch=0):
    return drop_based_on_prob(drop_probs(grad_sim, activation, kind), comp, u1, u2, branch)


def perform_dropout(feats, level, scores, comp, uniforms, branches=None):
    """FilterDropout.py:45-89.  feats: list of [B, C, H, W]; uniforms[idx] = (u1, u2), each [U, C]; branches[idx] =
    the random.randint(0, 1) of that level (score-driven complementary masks only)."""
    f1, f2 = [], []
    for idx, feat in enumerate(feats):
        bs = feat.shape[0]
        unlab = feat[bs // 2:]
        if idx in level:
            u1, u2 = uniforms[idx]
            if scores is None or bool(torch.all(scores[idx].eq(0))):
                if scores is None and comp:
                    m1 = (u1 < 0.5).float() * 2.0                      # Binomial(0.5).sample * 2
                    m2 = 2.0 - m1
                else:                                                  # two independent nn.Dropout2d(0.5)
                    m1 = (u1 < 0.5).float() * 2.0
                    m2 = (u2 < 0.5).float() * 2.0
                m1, m2 = m1[..., None, None], m2[..., None, None]
            else:
                act = unlab.detach().mean(dim=(2, 3))                            # adaptive_avg_pool2d(unlab_feat, (1, 1))
                m1, m2 = scores_dropout_v2(scores[idx], act, comp, "sigmoid", u1, u2, 0 if branches is None else branches[idx])
            p1, p2 = m1 * unlab, m2 * unlab
        else:
            p1, p2 = unlab, unlab
        f1.append(torch.cat((feat, p1)))
        f2.append(torch.cat((feat, p2)))
    return f1, f2


def dual_decoder_2d_dropout(sd, x, level, scores, comp, uniforms, branches=None, train=True, drop=None, update_stats=True):
    """DualDecoder.forward(x, dropout=True, ...) (unet.py:277-285): logits of (B + U) samples from both decoders."""
    ctx = nets.Ctx(train, drop, update_stats)
    feats = nets.encoder_2d(sd, x, ctx)
    f1, f2 = perform_dropout(feats, level, scores, comp, uniforms, branches)
    return nets.decoder_2d(sd, "decoder1", f1, ctx), nets.decoder_2d(sd, "decoder2", f2, ctx)
