"""Where the stochastic victims get the start index of their farthest-point sampling.

PointNet++ (model/pointnet2_utils.py:75 of the reference) and PCT (util/other_utils.py:264) draw
``torch.randint(0, N, (B,))`` from the global CPU generator in EVERY forward pass and move it to the GPU.  That
host-to-device copy is a synchronisation, so a loop around such a victim cannot be captured into a hipGraph.

The draws do not depend on any data, only on (B, N) and on how many came before.  An attack that knows how many
forward passes it will run can therefore take all of them from the CPU generator up front -- in the reference's order,
so a seeded run still follows the reference -- upload them once, and let the victim read "the row of this forward
pass" from device memory:

* ``LiveStarts``   the default: draw now, copy now (the reference's behaviour; not capturable);
* ``StartFeed``    ``forwards x len(highs)`` pre-drawn index vectors in HBM plus a device-side cursor; ``next()`` is an
                   ``index_select`` on the cursor and the last call of a forward pass advances it with an in-place add,
                   so both are recorded by a capture and every replay reads the next row.

A victim advertises its draws with ``fps_start_plan(N) -> [high, ...]`` (one entry per FPS call of a forward pass, in
call order); ``feed_for(model, ...)`` builds the feed, ``using(feed)`` makes it the source for the calls inside.
"""
import contextlib

import torch


class LiveStarts:
    def next(self, B, N, device):
        return torch.randint(0, N, (B,), dtype=torch.long).to(device)


class StartFeed:
    def __init__(self, highs, B, forwards, device, table=None):
        self.highs, self.B, self.forwards = list(highs), B, forwards
        if table is None:  # global CPU generator, call order: forward by forward, FPS call by FPS call
            table = torch.stack([torch.stack([torch.randint(0, h, (B,), dtype=torch.long) for h in self.highs])
                                 for _ in range(forwards)]) if forwards else torch.zeros(0, len(self.highs), B, dtype=torch.long)
        self.table = table.to(device)  # [forwards, calls, B]
        self.cursor = torch.zeros(1, dtype=torch.long, device=device)
        self.call = 0

    @staticmethod
    def draw(highs, B, forwards):
        """The CPU table alone (for callers that interleave these draws with others of their own)."""
        return torch.stack([torch.stack([torch.randint(0, h, (B,), dtype=torch.long) for h in highs])
                            for _ in range(forwards)])

    @classmethod
    def empty(cls, highs, B, capacity, device):
        """Room for ``capacity`` forward passes, every start 0 until ``load`` fills a range (an attack whose own draws
        interleave with the victim's fills it segment by segment, in the reference's order)."""
        return cls(highs, B, capacity, device, table=torch.zeros(capacity, len(highs), B, dtype=torch.long))

    def load(self, offset, forwards):
        """Draw ``forwards`` passes NOW from the CPU generator into rows [offset, offset + forwards) and point the
        cursor at the first of them (the addresses a captured graph holds stay the same)."""
        if forwards > 0:
            self.table[offset:offset + forwards].copy_(self.draw(self.highs, self.B, forwards))
        self.seek(offset)

    def seek(self, forward):
        """Next forward pass reads row ``forward`` (a device fill: allowed between replays, not inside a capture)."""
        self.cursor.fill_(forward)
        self.call = 0

    def next(self, B, N, device):
        j = self.call
        if B != self.B or N != self.highs[j]:
            raise RuntimeError("FPS start feed was drawn for (B=%d, N=%s), the victim asks for (B=%d, N=%d) at call %d"
                               % (self.B, self.highs, B, N, j))
        row = self.table[:, j].index_select(0, self.cursor)[0]
        self.call = (j + 1) % len(self.highs)
        if self.call == 0:
            self.cursor.add_(1)
        return row


_active = LiveStarts()


def next_start(B, N, device):
    """The start indices of one FPS call, int64 [B] on ``device``."""
    return _active.next(B, N, device)


@contextlib.contextmanager
def using(feed):
    """Victim forwards inside draw from ``feed`` (None: leave the current source in place)."""
    global _active
    if feed is None:
        yield
        return
    prev, _active = _active, feed
    try:
        yield
    finally:
        _active = prev


def plan_of(model, N):
    """The victim's FPS draws per forward pass ([] for a deterministic victim)."""
    plan = getattr(model, 'fps_start_plan', None)
    return list(plan(N)) if plan is not None else []


def feed_for(model, B, N, forwards, device):
    """A StartFeed for ``forwards`` passes of ``model`` on [B,*,N] inputs, drawn NOW from the CPU generator; None for a
    victim that draws nothing."""
    highs = plan_of(model, N)
    return StartFeed(highs, B, forwards, device) if highs else None
