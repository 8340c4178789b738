"""Host cost of RecognitionPredictor.generate's device loop (stop rules, token bookkeeping, scheduling) per emitted token, measured on
the CPU against a fake model whose calls return immediately (vectorised scripted tokens): what the Python side adds per decode call
when the device is infinitely fast. If this exceeds the device's time per call (4 steps x ~1.2 ms), the GPU idles.

    python tools/hostbench/loop_host_cost.py [--lines 2842] [--slots 256] [--max-tokens 48]
"""
import argparse, os, sys, time
from collections import deque
from types import SimpleNamespace
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from surya_amd.settings import settings

EOS, PAD, NOP = 1, 0, 3


class FastFake:
    def __init__(self, max_slots, stop_at):
        self.max_slots = max_slots
        self.c = SimpleNamespace(max_prefill_tokens=10 ** 6, max_slots=max_slots)
        self.cfg = SimpleNamespace(encoder=SimpleNamespace(spatial_merge_size=2))
        self.line = np.full(max_slots, -1, np.int64); self.pos = np.zeros(max_slots, np.int64)
        self.active = np.zeros(0, np.int64)
        self.stop_at = stop_at
        self.q = deque()
        self.calls = 0

    def _tok(self, line, pos):
        t = 100 + (line * 37 + pos * 11) % 9000
        return np.where(pos >= self.stop_at[line], EOS, t).astype(np.int32)

    def encode_ahead(self, tiles, grid_hw): pass

    def prefill(self, tiles, grid_hw, input_ids, slot_ids):
        s = np.asarray(slot_ids); ln = np.asarray([ids[0] - 1000 for ids in input_ids])
        self.line[s] = ln; self.pos[s] = 1
        tok = np.zeros((1, self.max_slots), np.int32); tok[0, s] = self._tok(ln, np.zeros_like(ln))
        self.out = (tok, np.full((1, self.max_slots), 0.5, np.float32), np.zeros((1, self.max_slots, 6), np.int32))

    def read_outputs(self, n): return self.out

    def set_active(self, slots): self.active = np.asarray(slots, np.int64)

    def decode_async(self, n, ring):
        tok = np.full((n, self.max_slots), -7, np.int32)
        a = self.active
        for k in range(n):
            tok[k, a] = self._tok(self.line[a], self.pos[a]); self.pos[a] += 1
        self.q.append((tok, np.full((n, self.max_slots), 0.25, np.float32), np.zeros((n, self.max_slots, 6), np.int32)))
        self.calls += 1

    def wait_outputs(self, n, ring): return self.q.popleft()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=2842); ap.add_argument("--slots", type=int, default=256)
    ap.add_argument("--max-tokens", type=int, default=48); ap.add_argument("--ragged", action="store_true")
    a = ap.parse_args()
    from surya_amd.recognition.predictor import RecognitionPredictor, RecognitionPrompt
    rng = np.random.default_rng(0)
    stop_at = rng.integers(20, 70, size=a.lines) if a.ragged else np.full(a.lines, 10 ** 9)
    pred = object.__new__(RecognitionPredictor)
    pred.prompt_queue, pred.batch_prompt_mapping = deque(), None
    pred.model = FastFake(a.slots, stop_at)
    pred.processor = SimpleNamespace(eos_token_id=EOS, pad_token_id=PAD, no_output_token=NOP)
    grids = [(2, 4)] * a.lines
    offs = np.cumsum([0] + [8] * a.lines)
    prep = {"prompts": [RecognitionPrompt(i, "ocr_with_boxes", None, None, True) for i in range(a.lines)],
            "max_tokens": {i: a.max_tokens for i in range(a.lines)}, "tiles": np.zeros((offs[-1], 3), np.float32), "tile_offs": offs,
            "grids": grids, "prompt_ids": [[1000 + i, 5, 6] for i in range(a.lines)]}
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        toks, _, _ = pred.generate(prep, a.slots)
        best = min(best, time.perf_counter() - t0)
    n_tok = sum(len(t) for t in toks)
    print(f"{a.lines} lines, {a.slots} slots, {n_tok} tokens, {pred.model.calls // 3} decode calls/pass: host loop {best * 1e3:.1f} ms = "
          f"{best / n_tok * 1e6:.2f} us/token, {best / (pred.model.calls / 3) * 1e3:.2f} ms per decode call (steps_per_sync {settings.RECOGNITION_STEPS_PER_SYNC})")


if __name__ == "__main__":
    main()
