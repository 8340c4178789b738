"""Host cost of assembling one recognised line (token stream -> TextLine with TextChars): the work bench.py's e2e leg does on the
worker thread beside the device loop. Random UTF-16 streams of `--tokens` ids, `--lines` lines; optional cProfile.

    python tools/hostbench/assemble_cost.py [--lines 2842] [--tokens 45] [--profile]
"""
import argparse, os, sys, time
from types import SimpleNamespace
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=2842); ap.add_argument("--tokens", type=int, default=45)
    ap.add_argument("--profile", action="store_true"); ap.add_argument("--words", action="store_true")
    ap.add_argument("--batch", type=int, default=64, help="lines per _assemble_batch call (1 = the per-line path)")
    ap.add_argument("--mixed", action="store_true", help="random ids over the whole vocabulary (what random weights emit) instead of text")
    a = ap.parse_args()
    from surya_amd.common.predictor import gc_paused
    from surya_amd.recognition.predictor import RecognitionPredictor
    from surya_amd.recognition.processor import SuryaOCRProcessor
    from surya_amd.recognition.tokenizer import ByteMathTokenizer, OCRTokenizer
    rng = np.random.default_rng(0)
    tok = OCRTokenizer(None, ByteMathTokenizer(256), reserve_special=64)
    proc = SuryaOCRProcessor(tok)
    pred = object.__new__(RecognitionPredictor)
    pred.processor = proc
    s_off = tok.special_token_offset
    n = a.lines
    flat = {"polygons": [[[10, 20], [500, 20], [500, 60], [10, 60]]] * n, "res_scales": [(1.0, 1.0)] * n,
            "slices": [np.zeros((40, 490, 3), np.uint8)] * n}
    if a.mixed:
        streams = [[int(t) for t in rng.integers(20, s_off + 0xd000, size=a.tokens)] for _ in range(n)]
    else:
        streams = [(s_off + rng.integers(0x20, 0x7f, size=a.tokens)).tolist() for _ in range(n)]
    scores = [rng.random(a.tokens).tolist() for _ in range(n)]
    boxes = [np.sort(rng.integers(0, 1025, size=(a.tokens, 6)), axis=0).astype(np.float32) for _ in range(n)]

    def run():
        with gc_paused():
            if a.batch <= 1:
                return [pred._assemble_line(flat, k, k, streams[k], scores[k], boxes[k], False, a.words, 1025) for k in range(n)]
            out = []
            for s0 in range(0, n, a.batch):
                out += pred._assemble_batch(flat, [(k, k, streams[k], scores[k], boxes[k]) for k in range(s0, min(n, s0 + a.batch))],
                                            False, a.words, 1025)
            return out

    run()
    if a.profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    best = min((lambda t0: (run(), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(3))
    print(f"{n} lines x {a.tokens} tokens: {best * 1e3:.1f} ms = {best / n * 1e6:.1f} us/line, {best / (n * a.tokens) * 1e6:.2f} us/char")


if __name__ == "__main__":
    main()
