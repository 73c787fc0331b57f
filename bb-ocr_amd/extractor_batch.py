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


def _ocr_input(image_path, image_index=None, decode_once=True):
    """``ocr_input_image`` for the batching loop: ``("ycc", triples, None)`` when the file easyocr would be given is a YCbCr-coded JPEG
    (decoded once, RGB + Y plane derived on the card: reader.decode_file_ycc) -- every thumbnail is, it is written as one -- else
    ``("rgb", rgb, gray)``."""
    from PIL import Image

    from .reader import decode_file, decode_file_ycc, reformat_input

    cover = image_index is None or image_index == 0
    max_dim = 1600 if cover else 2400
    try:
        img = Image.open(image_path)
        if max(img.size) > max_dim:
            img = img.convert("RGB")
            img.thumbnail((max_dim, max_dim))
            buf = io.BytesIO()
            img.save(buf, format="JPEG", quality=(90 if cover else 95))
            data = buf.getvalue()
            ycc = decode_file_ycc(data, padded=True) if decode_once else None
            return ("ycc", ycc, None) if ycc is not None else ("rgb",) + tuple(decode_file(data))
    except Exception:
        pass                                                      # :511-514: any failure falls back to the original file
    ycc = decode_file_ycc(os.fspath(image_path), padded=True) if decode_once else None
    return ("ycc", ycc, None) if ycc is not None else ("rgb",) + tuple(reformat_input(os.fspath(image_path)))


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


def extract_texts(reader, image_paths, ocr_image_indices=None, max_batch=64, decode_workers=None, decode_once=True, **readtext_kw):
    """``{index: text}`` for every index of ``ocr_image_indices`` (default: all pages), text = ``" ".join(r[1] for r in results)``
    exactly as :521; a page whose OCR fails gets ``""`` like :529-531.  Pages of equal (down-scaled) shape travel in one device
    batch of at most ``max_batch`` pages (``read_files`` with the reference's OCR-input rule as the decode step)."""
    res = read_files(reader, image_paths, ocr_image_indices, max_batch, decode_workers,
                     decode=lambda path, i: _ocr_input(path, i, decode_once), **readtext_kw)
    return {i: " ".join(t[1] for t in r) for i, r in res.items()}


def _plain_input(path, i=None):
    """Decode step of ``read_files`` without the extractor's thumbnail rule: what ``Reader.readtext(path)`` would hold."""
    from .reader import decode_file_ycc, reformat_input

    ycc = decode_file_ycc(os.fspath(path), padded=True)
    return ("ycc", ycc, None) if ycc is not None else ("rgb",) + tuple(reformat_input(os.fspath(path)))


def read_files(reader, image_paths, indices=None, max_batch=64, decode_workers=None, decode=_plain_input, **readtext_kw):
    """``{index: readtext result}`` for the files ``image_paths[i]``, i in ``indices`` (default: all): the result lists
    ``Reader.readtext(path)`` returns page by page, from 64-page device batches.  A page whose decode or OCR fails maps to ``[]``.

    Decoding (and the thumbnail + JPEG round trip) is what bounds the application once the OCR itself runs at hundreds of pages
    per second: a 1280x960 JPEG costs ~9 ms of one core decoded twice (RGB and the Y plane), ~4.6 ms decoded once into YCbCr triples
    (reader.decode_file_ycc: both planes are then derived on the card).  The files are therefore decoded by ``decode_workers`` threads (default:
    the host's cores, at most 16; PIL releases the GIL while decoding) and a shape group is sent to the device as soon as it is
    full, so the decode of later pages overlaps the device batch of earlier ones (ctypes releases the GIL during the C call).
    ``decode(path, index)`` returns ``("ycc", triples, None)`` or ``("rgb", rgb, gray)``."""
    import collections
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor

    if indices is None:
        indices = range(len(image_paths))
    idxs = [i for i in indices if 0 <= i < len(image_paths)]
    texts = {i: [] for i in idxs}
    if not idxs:
        return texts
    if decode_workers is None:
        # half of the process's CPU share (affinity mask and cgroup quota, bbocr_host_cpu_share), at most 8: the library's own host pool, the
        # upload stage and the two device-call threads need the rest (16-CPU share: 4 threads 700-780 pages/s, 8: 720-790, 16: 755-825)
        try:
            share = int(reader._lib.bbocr_host_cpu_share())
        except Exception:
            share = os.cpu_count() or 1
        decode_workers = max(1, min(8, share // 2, len(idxs)))

    # three overlapped stages: decode pool -> assembler thread (groups pages by shape, hands a full group on as one batch)
    # -> this thread (device call + result strings).  Back-pressure end to end: at most `window` decoded pages exist outside the two
    # assembled batches the queue may hold (a decode is only submitted once a slot is free, and a slot is released when its page has
    # been copied into a batch), so the resident set is bounded by ~4 batches however many files are queued.
    batches = queue.Queue(maxsize=2)
    window = max(2 * max_batch, 2 * decode_workers)
    slots = threading.Semaphore(window)

    def decode_one(i):
        try:
            return decode(image_paths[i], i)
        except Exception:
            return None

    def assemble():
        try:
            by_shape = {}

            def flush(group):
                # pages travel as LISTS: the Reader uploads them one by one into the device batch (no 236-MB np.stack on this thread)
                gray = None if group[0][2] is None else [p[2] for p in group]                 # None: a group of once-decoded YCbCr pages
                batches.put(([p[0] for p in group], [p[1] for p in group], gray))
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
                        pending.append((i, pool.submit(decode_one, i)))
                    if not pending:
                        continue
                    i, fut = pending.popleft()
                    page = fut.result()
                    del fut                                         # the future would keep the decoded arrays alive
                    if page is None:
                        slots.release()
                        continue
                    kind, rgb, gray = page                          # kind "ycc": rgb holds the YCbCr triples, gray is None
                    key = (kind, rgb.shape)
                    group = by_shape.setdefault(key, [])
                    group.append((i, rgb, gray))
                    if len(group) >= max_batch:
                        flush(by_shape.pop(key))
                    elif not pending and not done_submitting:
                        # nothing is being decoded: the next submission needs a slot.  If every slot is held by pages waiting in partial
                        # groups (many distinct shapes), send the largest group -- only then: while decodes are pending their pages hold
                        # slots too, and flushing on that account cut full batches into single pages (round 3: the faster the decode
                        # pool, the smaller the device batches)
                        if slots.acquire(blocking=False):
                            slots.release()
                        else:
                            big = max(by_shape, key=lambda k: len(by_shape[k]))
                            flush(by_shape.pop(big))
                for group in by_shape.values():
                    if group:
                        flush(group)
        finally:
            batches.put(None)

    def upload(item):
        """Stage between the assembler and the device calls: the batch's pages reach the card (one GIL-free C call on the context's upload
        stream, outside the call slots) while both device workers are still inside their calls -- a worker that uploaded its own batch left
        the card idle for that long, and the two workers fell into step.  Readers without the upload entry (test doubles) pass through."""
        ids, rgb, gray = item
        to_dev = getattr(reader, "_to_dev", None)
        if to_dev is None:
            return item
        try:
            return ids, to_dev(rgb), (to_dev(gray) if gray is not None else None), rgb, gray
        except Exception:
            return item                                  # the device worker retries from the host pages and reports per page

    def ocr(item):
        if len(item) == 5:                               # uploaded: device tensors + the host pages for the page-by-page retry
            ids, rgb_dev, gray_dev, rgb, gray = item
            try:
                if gray is None:
                    res = reader.readtext_device(*reader.pages_from_ycc(rgb_dev), **readtext_kw)
                else:
                    res = reader.readtext_device(rgb_dev, gray_dev, **readtext_kw)
                for i, r in zip(ids, res):
                    texts[i] = r
                return
            except Exception:
                pass
            del rgb_dev, gray_dev
        else:
            ids, rgb, gray = item
        read = (lambda a, g: reader.readtext_ycc_arrays(a, **readtext_kw)) if gray is None else (lambda a, g: reader.readtext_arrays(a, g, **readtext_kw))
        try:
            res = read(rgb, gray)
        except Exception:
            # the reference loses ONE page when its OCR fails (enhanced_extractor.py:529-531): retry the batch page by page
            res = []
            for k in range(len(ids)):
                try:
                    res.append(read(rgb[k:k + 1], None if gray is None else gray[k:k + 1])[0])
                except Exception:
                    res.append([])
        for i, r in zip(ids, res):
            texts[i] = r

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
            item = upload(item)                          # this thread is the upload stage: it runs one batch ahead of the device workers
            in_flight.acquire()
            fut = device_pool.submit(ocr, item)
            fut.add_done_callback(lambda _f: in_flight.release())
            futs.append(fut)
            del item
        for fut in futs:
            fut.result()
    worker.join()
    return texts
