"""Seeded synthetic book pages (nothing can be downloaded, so bench and tests draw their own).

Dark glyph strokes (30 +- 10) on a light noisy background (235 + N(0,4)), rendered with PIL's
built-in scalable font without anti-aliasing, words laid out explicitly so that word gaps
(16 px) and line pitch (40 px) leave wide margins for the designed detector weights
(``weights.designed_craft_state``).
"""
from __future__ import annotations

import numpy as np

_WORDS = ("the quick brown fox jumps over lazy dog beyond frontier romance of early days in middle west chapter "
          "copyright published october great britain stone first novel harry potter and philosopher express land "
          "endurance incredible voyage red men iowa dreaming city previously edited version entitled book cover page "
          "author title publisher edition isbn year library press university history volume series paper print").split()


def page(seed: int, width: int = 1280, height: int = 960, lines: int = 20, font_size: int = 24, word_gap: int = 16, line_pitch: int = 40,
         margin: int = 48, colour: bool = False, vocab=None, faint: float = 0.0):
    """-> (rgb uint8 [H,W,3], word boxes [(x0,y0,x1,y1,text)]).

    ``colour=True`` tints paper and ink (a per-page offset on the red and blue channels plus per-pixel chroma noise; the green
    channel -- the one the designed detector reads -- keeps the grey page's values), so that R != G != B everywhere and the
    gray plane (cv2 BGR2GRAY on the device) is a genuine three-channel mix.
    ``vocab`` replaces the built-in word list (tests/golden/train_crnn.py mixes in random letter strings).
    ``faint`` (0..1): that share of the text LINES is printed in an ink whose luminance nearly equals the paper's -- the green channel
    (what the designed detector reads) keeps its dark strokes, red and blue are set so that cv2's gray plane shows the line at a
    contrast of a few grey levels under the page noise.  The recogniser's confidence on such lines falls below ``contrast_ths`` and
    upstream's contrast retry (``adjust_contrast_grey``) becomes live: the low-confidence workload of bench.py / the GPU tests."""
    from PIL import Image, ImageDraw, ImageFont

    rng = np.random.default_rng(seed)
    try:
        font = ImageFont.load_default(size=font_size)
    except TypeError:  # very old Pillow: bitmap font only
        font = ImageFont.load_default()
    mask = Image.new("L", (width, height), 0)
    draw = ImageDraw.Draw(mask)
    draw.fontmode = "1"
    words = []
    vocab = _WORDS if vocab is None else vocab
    y = margin
    faint_rows = []
    for _ in range(lines):
        if y + line_pitch > height - margin // 2:
            break
        if faint > 0 and rng.random() < faint:
            faint_rows.append((max(0, y - 6), min(height, y + line_pitch - 6)))
        x = margin + int(rng.integers(0, 24))
        limit = width - margin - int(rng.integers(0, width // 3))
        while True:
            w = vocab[int(rng.integers(0, len(vocab)))]
            if rng.random() < 0.15:
                w = w.capitalize()
            bb = draw.textbbox((x, y), w, font=font)
            if bb[2] > limit:
                break
            draw.text((x, y), w, fill=255, font=font)
            words.append((bb[0], bb[1], bb[2], bb[3], w))
            x = bb[2] + word_gap
        y += line_pitch
    m = np.asarray(mask) > 127
    bg = 235.0 + rng.normal(0.0, 4.0, (height, width))
    fg = 30.0 + rng.uniform(-10.0, 10.0, (height, width))
    g = np.where(m, fg, bg)
    g = np.clip(np.rint(g), 0, 255).astype(np.uint8)
    if faint_rows:
        # gray = (9798 R + 19235 G + 3735 B) >> 15: choose R = B = v so that ink (G ~ 30) and paper (G ~ 235) land `delta` grey levels apart
        frng = np.random.default_rng(seed + 11_000_027)
        rb = np.repeat(g[:, :, None], 2, axis=2).astype(np.float64)
        for (r0, r1) in faint_rows:
            delta = float(frng.uniform(4.0, 14.0))
            # paper: R = B = 60 -> gray ~ 0.413 * 60 + 0.587 * 235 = 162.7; ink: gray target = paper - delta
            v_ink = (162.7 - delta - 0.587 * g[r0:r1].astype(np.float64)) / 0.413
            v = np.where(m[r0:r1], v_ink, 60.0 + frng.normal(0.0, 2.0, (r1 - r0, width)))
            rb[r0:r1, :, 0] = v
            rb[r0:r1, :, 1] = v
        rb = np.clip(np.rint(rb), 0, 255).astype(np.uint8)
        return np.ascontiguousarray(np.stack([rb[..., 0], g, rb[..., 1]], axis=2)), words
    if not colour:
        return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2)), words
    crng = np.random.default_rng(seed + 7_000_003)          # own stream: the grey page of a seed does not change
    paper = crng.integers(-28, 17, 2)                        # (red, blue) offsets: cream / bluish / greenish stock
    ink = crng.integers(0, 60, 2)                            # dark blue / sepia ink
    off = np.where(m[:, :, None], ink[None, None, :], paper[None, None, :]) + crng.integers(-3, 4, (height, width, 2))
    rb = np.clip(g[:, :, None].astype(np.int64) + off, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.stack([rb[..., 0], g, rb[..., 1]], axis=2)), words


def pages(n: int, seed0: int = 1234, **kw):
    return np.stack([page(seed0 + i, **kw)[0] for i in range(n)])
