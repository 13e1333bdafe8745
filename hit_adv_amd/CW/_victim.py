"""The victim as the CW attacks call it.  A victim that offers ``attack_view()`` (PointNet: the HIP engine; DGCNN: the
folded EdgeConv view) is differentiated through that view -- the same function as the eval-mode module up to fp32
re-association, several times faster than the module's MIOpen / op-by-op formulation; its buffers are re-folded from the
module's current weights at the start of every ``attack()``.  ``fast_victim=False`` in an attack's constructor keeps the
module itself (the reference's behaviour, ShapeAttack/HiT_ADV.py has the same switch)."""


import torch

from ..model import _sampling


class Victim:
    def __init__(self, model, fast=True):
        self.model, self.fast, self.view, self.feed = model, fast, None, None

    def open_feed(self, B, N, capacity, device):
        """A victim that draws FPS start indices from the CPU generator in every forward pass (PointNet++, PCT) reads
        them from a device-resident table inside a captured loop (model/_sampling.py); ``load`` fills it range by range
        at the points where the reference's forward passes would have drawn.  No-op for deterministic victims."""
        highs = _sampling.plan_of(self.model, N)
        self.feed = _sampling.StartFeed.empty(highs, B, max(2, capacity), device) if highs else None

    def load(self, offset, forwards):
        if self.feed is not None:
            self.feed.load(offset, forwards)

    def draw(self, forwards):
        """The CPU table of ``forwards`` passes' draws, taken NOW from the CPU generator (None for a deterministic victim):
        an attack that takes all its random numbers up front calls this where the reference's passes would have drawn and
        ``put``s the rows when their turn comes."""
        if self.feed is None or forwards <= 0:
            return None
        return _sampling.StartFeed.draw(self.feed.highs, self.feed.B, forwards)

    def put(self, offset, table):
        """Rows drawn earlier (``draw``) into [offset, offset + len) and the cursor onto the first of them."""
        if self.feed is not None:
            if table is not None:
                self.feed.table[offset:offset + table.shape[0]].copy_(table)
            self.feed.seek(offset)

    def put_all(self, tables):
        """Every row of the attack at once, in pass order (a list of ``draw`` results; None entries skipped): ONE upload at
        setup.  An upload from pageable host memory waits for the stream it is issued on -- inside the loop it would hold the
        host until the binary step before it has run, and with it every other attack the host has yet to queue."""
        if self.feed is not None:
            rows = [t for t in tables if t is not None]
            if rows:
                rows = torch.cat(rows)
                self.feed.table[:rows.shape[0]].copy_(rows)

    def seek(self, offset):
        if self.feed is not None:
            self.feed.seek(offset)

    def close_feed(self):
        """End of an attack's loop: drop the feed; loud if a fused fp16x2 layer of the victim left fp16's range meanwhile."""
        self.feed = None
        from ..model import _pointwise
        p = next(self.model.parameters(), None)
        if p is not None:
            _pointwise.check_range(p.device)
        if hasattr(self.view, 'check_range'):
            self.view.check_range()

    def prepare(self):
        if not (self.fast and hasattr(self.model, 'attack_view')):
            return
        try:
            if self.view is None:
                self.view = self.model.attack_view()
            else:
                self.view.refresh(self.model)
        except NotImplementedError:  # a configuration the view does not cover
            self.fast, self.view = False, None

    def __call__(self, x):
        with _sampling.using(self.feed):
            out = (self.view if self.view is not None else self.model)(x)
        return out[0] if isinstance(out, tuple) else out
