"""Caller-side batching for the reference's extractor.

``pipeline_demo/extractor/enhanced_extractor.py`` runs OCR page by page: the loop at :680-688 calls
``extract_text_with_ocr`` per image, which (after the optional crop logic) applies the down-scaling rule of :486-512 and then
``reader.readtext(path, paragraph=False, batch_size=1, workers=0)`` (:520), joins ``result[1]`` with spaces (:521) and maps any
exception to the empty string (:529-531).  ``extract_texts`` does the same for ALL pages of a book (or of many books) with ONE
``readtext_batched`` call per page shape, so the backend sees 64-page batches instead of single pages.

Only the OCR step is batched; cropping / LLM steps of the extractor stay where they are.  The down-scaling rule is restated
exactly, including its JPEG round trip (the reference writes the thumbnail as JPEG quality 90/95 and lets easyocr decode it).
"""
from __future__ import annotations

import io
import os

import numpy as np


def ocr_input_image(image_path, image_index=None):
    """The pixels ``extract_text_with_ocr`` hands to easyocr for ``image_path`` (enhanced_extractor.py:486-512): pages whose
    longer side exceeds 1600 px (cover, ``image_index`` None or 0) / 2400 px (other pages) are ``Image.thumbnail``-ed to that
    size in RGB and re-encoded as JPEG quality 90 / 95; everything else is read as is.  Returns what ``reformat_input`` would
    produce for the file easyocr is given: (RGB uint8 [H,W,3], gray uint8 [H,W])."""
    from PIL import Image

    from .reader import decode_file, reformat_input

    cover = image_index is None or image_index == 0
    max_dim = 1600 if cover else 2400
    try:
        img = Image.open(image_path)
        if max(img.size) > max_dim:
            img = img.convert("RGB")
            img.thumbnail((max_dim, max_dim))
            buf = io.BytesIO()
            img.save(buf, format="JPEG", quality=(90 if cover else 95))
            return decode_file(buf.getvalue())                    # what easyocr's loader sees: a JPEG file on disk
    except Exception:
        pass                                                      # :511-514: any failure falls back to the original file
    return reformat_input(os.fspath(image_path))


def extract_texts(reader, image_paths, ocr_image_indices=None, max_batch=64, decode_workers=None, **readtext_kw):
    """``{index: text}`` for every index of ``ocr_image_indices`` (default: all pages), text = ``" ".join(r[1] for r in results)``
    exactly as :521; a page whose OCR fails gets ``""`` like :529-531.  Pages of equal (down-scaled) shape travel in one device
    batch of at most ``max_batch`` pages.

    Decoding (and the thumbnail + JPEG round trip) is what bounds the application once the OCR itself runs at hundreds of pages
    per second: a 1280x960 JPEG costs ~10 ms of one core.  The files are therefore decoded by ``decode_workers`` threads (default:
    the host's cores, at most 16; PIL releases the GIL while decoding) and a shape group is sent to the device as soon as it is
    full, so the decode of later pages overlaps the device batch of earlier ones (ctypes releases the GIL during the C call)."""
    import collections
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor

    if ocr_image_indices is None:
        ocr_image_indices = range(len(image_paths))
    idxs = [i for i in ocr_image_indices if 0 <= i < len(image_paths)]
    texts = {i: "" for i in idxs}
    if not idxs:
        return texts
    if decode_workers is None:
        decode_workers = max(1, min(8, os.cpu_count() or 1, len(idxs)))

    # three overlapped stages: decode pool -> assembler thread (groups pages by shape, stacks a full group into one host batch)
    # -> this thread (device call + result strings).  Back-pressure end to end: at most `window` decoded pages exist outside the two
    # assembled batches the queue may hold (a decode is only submitted once a slot is free, and a slot is released when its page has
    # been copied into a batch), so the resident set is bounded by ~4 batches however many files are queued.
    batches = queue.Queue(maxsize=2)
    window = max(2 * max_batch, 2 * decode_workers)
    slots = threading.Semaphore(window)

    def decode(i):
        try:
            return ocr_input_image(image_paths[i], i)
        except Exception:
            return None

    def assemble():
        try:
            by_shape = {}

            def flush(group):
                batches.put(([p[0] for p in group], np.stack([p[1] for p in group]), np.stack([p[2] for p in group])))
                for _ in group:
                    slots.release()

            with ThreadPoolExecutor(max_workers=decode_workers) as pool:
                pending = collections.deque()
                it = iter(idxs)
                done_submitting = False
                while pending or not done_submitting:
                    while not done_submitting and len(pending) < window and slots.acquire(blocking=not pending):
                        i = next(it, None)
                        if i is None:
                            slots.release()
                            done_submitting = True
                            break
                        pending.append((i, pool.submit(decode, i)))
                    if not pending:
                        continue
                    i, fut = pending.popleft()
                    page = fut.result()
                    del fut                                         # the future would keep the decoded arrays alive
                    if page is None:
                        slots.release()
                        continue
                    rgb, gray = page
                    group = by_shape.setdefault(rgb.shape, [])
                    group.append((i, rgb, gray))
                    if len(group) >= max_batch:
                        flush(by_shape.pop(rgb.shape))
                    elif not slots.acquire(blocking=False):
                        # every slot is held by pages waiting in partial groups (many distinct shapes): send the largest one
                        big = max(by_shape, key=lambda k: len(by_shape[k]))
                        flush(by_shape.pop(big))
                    else:
                        slots.release()
                for group in by_shape.values():
                    if group:
                        flush(group)
        finally:
            batches.put(None)

    def ocr(item):
        ids, rgb, gray = item
        try:
            res = reader.readtext_arrays(rgb, gray, **readtext_kw)
        except Exception:
            # the reference loses ONE page when its OCR fails (enhanced_extractor.py:529-531): retry the batch page by page
            res = []
            for k in range(len(ids)):
                try:
                    res.append(reader.readtext_arrays(rgb[k:k + 1], gray[k:k + 1], **readtext_kw)[0])
                except Exception:
                    res.append([])
        for i, r in zip(ids, res):
            texts[i] = " ".join(t[1] for t in r)

    worker = threading.Thread(target=assemble, daemon=True)
    worker.start()
    # two device batches in flight on the one Reader (bbocr_config::call_slots, the reference's own ThreadPoolExecutor contract,
    # batch_processor_enhanced.py:215): the H2D copy and detector of batch k+1 run while batch k's host thread finishes its boxes and strings
    in_flight = threading.Semaphore(2)
    with ThreadPoolExecutor(max_workers=2, thread_name_prefix="bbocr-ocr") as device_pool:
        futs = []
        while True:
            item = batches.get()
            if item is None:
                break
            in_flight.acquire()
            fut = device_pool.submit(ocr, item)
            fut.add_done_callback(lambda _f: in_flight.release())
            futs.append(fut)
            del item
        for fut in futs:
            fut.result()
    worker.join()
    return texts
