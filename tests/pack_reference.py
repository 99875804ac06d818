"""Independent numpy statement of the packed weight stream (csrc/nerf_layout.h) used to check
nerf_pack_model.  Test infrastructure."""
import numpy as np


def act_feat(t, r, h):
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h


def pe_xyz_feat(n, h):
    if n < 30:
        a = 15 * h + n // 2
        return 3 + 6 * (a // 3) + 3 * (n & 1) + a % 3
    if n == 30:
        return 0 if h == 0 else 2
    return 1 if h == 0 else -1


def pe_dir_feat(n, h):
    if n < 12:
        a = 6 * h + n // 2
        return 3 + 6 * (a // 3) + 3 * (n & 1) + a % 3
    if n == 12:
        return 0 if h == 0 else 2
    if n == 13:
        return 1 if h == 0 else -1
    return -1


def _stream(W, ksteps, ntiles, colmap):
    """W [out,in]; colmap(s, h) -> input column or -1.  Returns the [ksteps/4, ntiles, 64, 4] stream."""
    out = np.zeros((ksteps // 4, ntiles, 64, 4), np.float32)
    for s in range(ksteps):
        g, q = divmod(s, 4)
        for h in range(2):
            c = colmap(s, h)
            if c < 0:
                continue
            for j in range(ntiles):
                out[g, j, 32 * h:32 * h + 32, q] = W[32 * j:32 * j + 32, c]
    return out.reshape(-1)


def pack_model(sd, prefix):
    g = lambda n: sd[f"{prefix}.{n}"].detach().cpu().numpy().astype(np.float32)
    hid = lambda s, h: act_feat(s >> 4, s & 15, h)
    parts = [_stream(g("pts_linears.0.weight"), 32, 8, pe_xyz_feat)]
    for i in (1, 2, 3, 4):
        parts.append(_stream(g(f"pts_linears.{i}.weight"), 128, 8, hid))
    W5 = g("pts_linears.5.weight")
    parts.append(_stream(W5, 32, 8, pe_xyz_feat))
    parts.append(_stream(W5, 128, 8, lambda s, h: 63 + hid(s, h)))
    for i in (6, 7):
        parts.append(_stream(g(f"pts_linears.{i}.weight"), 128, 8, hid))
    parts.append(_stream(g("feature_linear.weight"), 128, 8, hid))

    def views_col(s, h):
        if s < 128:
            return hid(s, h)
        c = pe_dir_feat(s - 128, h)
        return -1 if c < 0 else 256 + c
    parts.append(_stream(g("views_linears.0.weight"), 144, 4, views_col))

    def bias_block(b, ntiles):
        o = np.zeros((2, ntiles * 16), np.float32)
        for h in range(2):
            for j in range(ntiles):
                for r in range(16):
                    o[h, j * 16 + r] = b[act_feat(j, r, h)]
        return o.reshape(-1)
    for i in range(8):
        parts.append(bias_block(g(f"pts_linears.{i}.bias"), 8))
    parts.append(bias_block(g("feature_linear.bias"), 8))
    parts.append(bias_block(g("views_linears.0.bias"), 4))
    parts.append(bias_block(g("alpha_linear.weight")[0], 8))
    Wr = g("rgb_linear.weight")
    for c in range(3):
        parts.append(bias_block(Wr[c], 4))
    parts.append(g("rgb_linear.bias"))
    parts.append(g("alpha_linear.bias"))
    return np.concatenate(parts)


# ------------------------------------------------------------------ fp16 stream (nerf_layout.h)
def act16_feat(s, j, h):
    return 16 * s + (j & 3) + 8 * (j >> 2) + 4 * h


def _frag(W, rows, colmap):
    """One 1-KiB A fragment: lane l=(i,h), element j = W[rows[i]][colmap(j,h)] (fp16), zero if row/col < 0."""
    f = np.zeros((64, 8), np.float32)
    for h in range(2):
        for j in range(8):
            c = colmap(j, h)
            if c < 0:
                continue
            for i in range(32):
                if rows[i] >= 0:
                    f[32 * h + i, j] = W[rows[i], c]
    return f.reshape(-1)


def pack_model_f16(sd, prefix, split=False):
    """-> (const region float32 [4096], fragment stream float16 [1184*512]); with split=True the "f32x"
    stream [2368*512]: every fragment followed by its low part, w = w_h + 2^-11 w_l."""
    g = lambda n: sd[f"{prefix}.{n}"].detach().cpu().numpy().astype(np.float32)
    frags = []
    rows_of = lambda m: [32 * m + i for i in range(32)]
    W0, W5 = g("pts_linears.0.weight"), g("pts_linears.5.weight")
    for m in range(8):                                           # L0
        for s in range(4):
            frags.append(_frag(W0, rows_of(m), lambda j, h, s=s: pe_xyz_feat(8 * s + j, h)))
    hidden = lambda W: [frags.append(_frag(W, rows_of(m), lambda j, h, s=s: act16_feat(s, j, h)))
                        for m in range(8) for s in range(16)]
    for i in (1, 2, 3, 4):
        hidden(g(f"pts_linears.{i}.weight"))
    for m in range(8):                                           # L5: 4 PE + 16 hidden k-steps per m
        for s in range(4):
            frags.append(_frag(W5, rows_of(m), lambda j, h, s=s: pe_xyz_feat(8 * s + j, h)))
        for s in range(16):
            frags.append(_frag(W5, rows_of(m), lambda j, h, s=s: 63 + act16_feat(s, j, h)))
    for i in (6, 7):
        hidden(g(f"pts_linears.{i}.weight"))
    Wa = g("alpha_linear.weight")
    for s in range(16):                                          # sigma head: row 0 (ahead of the feature layer)
        frags.append(_frag(Wa, [0] + [-1] * 31, lambda j, h, s=s: act16_feat(s, j, h)))
    hidden(g("feature_linear.weight"))
    Wv = g("views_linears.0.weight")
    for m in range(4):                                           # views: 16 feature + 2 dir k-steps
        for s in range(16):
            frags.append(_frag(Wv, rows_of(m), lambda j, h, s=s: act16_feat(s, j, h)))
        for s in range(2):
            def col(j, h, s=s):
                c = pe_dir_feat(8 * s + j, h)
                return -1 if c < 0 else 256 + c
            frags.append(_frag(Wv, rows_of(m), col))
    Wr = g("rgb_linear.weight")
    for s in range(8):                                           # rgb head: rows 0..2
        frags.append(_frag(Wr, [0, 1, 2] + [-1] * 29, lambda j, h, s=s: act16_feat(s, j, h)))
    assert len(frags) == 1184
    if split:
        out = []
        for f32 in frags:
            hi = f32.astype(np.float16)
            lo = ((f32 - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
            out += [hi, lo]
        stream = np.concatenate(out)
    else:
        stream = np.concatenate(frags).astype(np.float16)

    const = np.zeros(4096, np.float32)

    def bias_block(b, ntiles):
        o = np.zeros((2, ntiles * 16), np.float32)
        for h in range(2):
            for j in range(ntiles):
                for r in range(16):
                    o[h, j * 16 + r] = b[act_feat(j, r, h)]
        return o.reshape(-1)
    for i in range(8):
        const[i * 256:(i + 1) * 256] = bias_block(g(f"pts_linears.{i}.bias"), 8)
    const[2048:2304] = bias_block(g("feature_linear.bias"), 8)
    const[2304:2432] = bias_block(g("views_linears.0.bias"), 4)
    const[2432:2435] = g("rgb_linear.bias")
    const[2435] = g("alpha_linear.bias")[0]
    return const, stream


# ------------------------------------------------------------------ f16s stream: 16x16x32 MFMA tiling (nerf_layout.h "f16s")
def act16s_feat(s, j, g):
    return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3)


def pe16s_xyz_feat(n, g):
    i, sc = n >> 1, n & 1
    if i < 6:
        return 3 + 6 * (2 * g + i // 3) + 3 * sc + i % 3
    x = 2 * g + (i - 6)
    if x < 6:
        return 3 + 6 * (8 + x // 3) + 3 * sc + x % 3
    if x == 6:
        return sc
    return 2 if sc == 0 else -1


def pe16s_dir_feat(n, g):
    i, sc = n >> 1, n & 1
    if i < 3:
        return 3 + 6 * g + 3 * sc + i
    if g == 0:
        return sc
    if g == 1:
        return 2 if sc == 0 else -1
    return -1


def _frag_s(W, rows, colmap):
    """One 1-KiB A fragment of the 16x16x32 tiling: lane l=(i,g), element j = W[rows[i]][colmap(j,g)], zero if row/col < 0."""
    f = np.zeros((64, 8), np.float32)
    for g in range(4):
        for j in range(8):
            c = colmap(j, g)
            if c < 0:
                continue
            for i in range(16):
                if rows[i] >= 0:
                    f[16 * g + i, j] = W[rows[i], c]
    return f.reshape(-1)


def pack_model_f16s(sd, prefix):
    """-> (const region float32 [4096]: biases in natural order, fragment stream float16 [1184*512]: 1172 fragments + 12 zero pads)."""
    g = lambda n: sd[f"{prefix}.{n}"].detach().cpu().numpy().astype(np.float32)
    frags = []
    rows_of = lambda m: [16 * m + i for i in range(16)]
    W0, W5 = g("pts_linears.0.weight"), g("pts_linears.5.weight")
    for m in range(16):                                          # L0
        for s in range(2):
            frags.append(_frag_s(W0, rows_of(m), lambda j, gg, s=s: pe16s_xyz_feat(8 * s + j, gg)))
    hidden = lambda W: [frags.append(_frag_s(W, rows_of(m), lambda j, gg, s=s: act16s_feat(s, j, gg)))
                        for m in range(16) for s in range(8)]
    for i in (1, 2, 3, 4):
        hidden(g(f"pts_linears.{i}.weight"))
    for m in range(16):                                          # L5: 2 PE + 8 hidden k-steps per m
        for s in range(2):
            frags.append(_frag_s(W5, rows_of(m), lambda j, gg, s=s: pe16s_xyz_feat(8 * s + j, gg)))
        for s in range(8):
            frags.append(_frag_s(W5, rows_of(m), lambda j, gg, s=s: 63 + act16s_feat(s, j, gg)))
    for i in (6, 7):
        hidden(g(f"pts_linears.{i}.weight"))
    Wa = g("alpha_linear.weight")
    for s in range(8):                                           # sigma head: row 0
        frags.append(_frag_s(Wa, [0] + [-1] * 15, lambda j, gg, s=s: act16s_feat(s, j, gg)))
    hidden(g("feature_linear.weight"))
    Wv = g("views_linears.0.weight")
    for m in range(8):                                           # views: 8 feature + 1 dir k-step
        for s in range(8):
            frags.append(_frag_s(Wv, rows_of(m), lambda j, gg, s=s: act16s_feat(s, j, gg)))

        def col(j, gg):
            c = pe16s_dir_feat(j, gg)
            return -1 if c < 0 else 256 + c
        frags.append(_frag_s(Wv, rows_of(m), col))
    Wr = g("rgb_linear.weight")
    for s in range(4):                                           # rgb head: rows 0..2
        frags.append(_frag_s(Wr, [0, 1, 2] + [-1] * 13, lambda j, gg, s=s: act16s_feat(s, j, gg)))
    assert len(frags) == 1172
    frags += [np.zeros(512, np.float32)] * 12
    stream = np.concatenate(frags).astype(np.float16)
    const = np.zeros(4096, np.float32)
    for i in range(8):
        const[i * 256:(i + 1) * 256] = g(f"pts_linears.{i}.bias")
    const[2048:2304] = g("feature_linear.bias")
    const[2304:2432] = g("views_linears.0.bias")
    const[2432:2435] = g("rgb_linear.bias")
    const[2435] = g("alpha_linear.bias")[0]
    return const, stream
