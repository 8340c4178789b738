"""Test infrastructure: a stand-in for HipLayoutModel (table family) that computes with oracle/layout_oracle.py on the CPU, so that
surya_amd.table_rec.TableRecPredictor's host logic can be driven without a GPU (live test against the reference's predictor) and so
that the GPU predictor test has a same-interface checker. Never imported by the product."""
import numpy as np
import torch

from oracle import layout_oracle as lo


class OracleTableModel:
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
