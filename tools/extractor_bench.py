"""Application-level rate of SURVEY 8 row f3: a directory of JPEG pages -> extractor_batch.extract_texts -> {index: text}, i.e. file read +
JPEG decode (thread pool) + H2D + the whole OCR path + Python results.  Usage: extractor_bench.py [n_pages] [decode_workers]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import bb_ocr_amd
from bb_ocr_amd import extractor_batch, synth, weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
workers = int(sys.argv[2]) if len(sys.argv) > 2 else None
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
with tempfile.TemporaryDirectory() as d:
    uniq = [synth.page(1234 + i)[0] for i in range(8)]
    paths = []
    for i in range(n):
        p = os.path.join(d, f"page_{i:04d}.jpg")
        Image.fromarray(uniq[i % 8]).save(p, quality=92)
        paths.append(p)
    t0 = time.perf_counter()
    for p in paths[:16]:
        np.asarray(Image.open(p).convert("RGB"))
    dec = (time.perf_counter() - t0) / 16
    extractor_batch.extract_texts(r, paths[:64], decode_workers=workers)          # warm-up
    bb_ocr_amd.freeze_gc()
    ref = None
    for w in ([workers] if workers else [1, 4, 8, 16]):
        for once in (False, True, False, True):            # each arm twice, alternating
            t0 = time.perf_counter()
            texts = extractor_batch.extract_texts(r, paths, decode_workers=w, decode_once=once)
            dt = time.perf_counter() - t0
            ref = ref or texts
            print(f"{n} JPEG pages 1280x960, {w} decode threads, {'ONE YCbCr decode per page' if once else 'two decodes per page (RGB + Y)'}: "
                  f"{dt*1e3:.0f} ms = {n/dt:.1f} pages/s (single-thread RGB decode {dec*1e3:.1f} ms/page; cores {os.cpu_count()}; "
                  f"non-empty texts {sum(bool(t) for t in texts.values())}; texts equal to the first run's: {texts == ref})", flush=True)
    # host arrays in (no decode): the PCIe-inclusive rate of the readtext boundary
    host = np.stack([uniq[i % 8] for i in range(64)])
    r.readtext_arrays(host)
    t0 = time.perf_counter()
    for _ in range(4):
        r.readtext_arrays(host)
    dt = (time.perf_counter() - t0) / 4
    print(f"readtext_arrays(host uint8 [64,960,1280,3]): {dt*1e3:.1f} ms per 64 pages = {64/dt:.1f} pages/s (H2D from pageable memory included)")
