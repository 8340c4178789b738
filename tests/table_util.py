"""Test infrastructure: a stand-in for HipLayoutModel (table family) that computes with oracle/layout_oracle.py on the CPU, so that
surya_amd.table_rec.TableRecPredictor's host logic can be driven without a GPU (live test against the reference's predictor) and so
that the GPU predictor test has a same-interface checker. Never imported by the product."""
import numpy as np
import torch

from oracle import layout_oracle as lo


class OracleFedRunsMixin:
    """surya_layout_set_feedback / _decode_steps / _wait_steps on top of a stand-in's decode_step, the fed-back token formed by the
    oracle's restatement of the reference loops (layout_oracle.fed_token_*): the CPU checker of the device-fed runs."""
    is_table = True
    dtype = torch.float32

    def set_feedback(self, page_sizes=None):
        self._sizes = None if page_sizes is None else [tuple(int(v) for v in r) for r in np.asarray(page_sizes).reshape(-1, 2)]
        self._fed, self._rings = None, {}

    def _fed_token(self, cls, box):
        d = self.cfg.decoder
        rows = []
        for j in range(cls.shape[0]):
            c, b = torch.from_numpy(cls[j]).to(self.dtype), torch.from_numpy(box[j]).to(self.dtype)
            if self.is_table:
                rows.append(lo.fed_token_table(c, b, d))
            else:
                rows.append(lo.fed_token_layout(c, b, d, None if self._sizes is None else self._sizes[j]))
        return torch.stack(rows).numpy().astype(np.int32)

    def decode_steps(self, boxes, position, n_steps, ring=0):
        assert 1 <= n_steps <= 16 and ring in (0, 1)
        tok = np.asarray(boxes, np.int32) if boxes is not None else self._fed
        assert tok is not None, "nothing to continue from"
        rec = []
        for k in range(n_steps):
            cls, box = self.decode_step(tok, position + k)
            tok = self._fed_token(cls, box)
            rec.append((cls, box, tok))
        self._fed = tok
        self._rings[ring] = rec

    def wait_steps(self, n_steps, ring=0):
        rec = self._rings.pop(ring)
        assert len(rec) == n_steps
        return np.stack([r[0] for r in rec]), np.stack([r[1] for r in rec]), np.stack([r[2] for r in rec])


class OracleTableModel(OracleFedRunsMixin):
    def __init__(self, cfg, sd, max_batch=32, max_boxes=1024):
        self.cfg, self.sd = cfg, sd
        self.max_batch, self.max_boxes = max_batch, max_boxes
        self.device = torch.device("cpu")
        self.enc = None
        self.map = None
        self.state = None

    @property
    def config(self):
        return self.cfg

    def encode_host(self, px):
        with torch.inference_mode():
            self.enc = lo.encoder_forward(self.sd, self.cfg.encoder, px.float())
        self.select(list(range(px.shape[0])))

    def select(self, src_index):
        idx = list(src_index)
        assert 0 < len(idx) <= self.max_batch and all(0 <= i < self.enc.shape[0] for i in idx)
        self.map = idx
        self.state = None

    def prefill(self, boxes):
        d = self.cfg.decoder
        self.state = lo.LayoutDecoderState(d.num_hidden_layers)
        b = torch.from_numpy(np.asarray(boxes, np.int64)).view(len(self.map), -1, 10)
        with torch.inference_mode():
            box, props = lo.decoder_forward(self.sd, d, b, self.enc[self.map], 0, self.state)
        cls = torch.cat([props[k][:, -1] for k, _ in d.head_widths() if k != "bbox"], -1)
        return cls.numpy().copy(), box[:, -1].numpy().copy()

    def decode_step(self, boxes, position):
        d = self.cfg.decoder
        if position == 0:
            self.state = lo.LayoutDecoderState(d.num_hidden_layers)
        b = torch.from_numpy(np.asarray(boxes, np.int64)).view(len(self.map), 1, 10)
        with torch.inference_mode():
            box, props = lo.decoder_forward(self.sd, d, b, self.enc[self.map], position, self.state)
        cls = torch.cat([props[k][:, -1] for k, _ in d.head_widths() if k != "bbox"], -1)
        return cls.numpy().copy(), box[:, -1].numpy().copy()
