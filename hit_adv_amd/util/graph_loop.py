"""Replaying one attack iteration as a hipGraph.

The CW-family attacks (and HiT-ADV, which has its own copy of this logic around its workspaces) repeat one body --
victim forward / backward, losses, Adam, clip, best-so-far bookkeeping -- hundreds to thousands of times on tensors
whose addresses do not change.  Run eagerly, such a loop is bound by the host (~100 kernel launches plus autograd and
optimizer bookkeeping per iteration); captured once and replayed it is bound by the GPU.

``IterationGraph(body, use_graph)``:
  * ``probe()`` runs ``body`` twice on a private stream, the second time under PyTorch's sync-debug mode set to
    "error": a body that would synchronise with the host (a victim drawing a CPU ``randint``, a loss calling ``.item()``)
    raises HERE, in eager mode, instead of invalidating a capture.  The CPU generator's state is restored afterwards, so
    the two extra passes do not move the attack's random draws.  The caller re-initialises its state tensors after it.
  * ``capture()`` records one ``body`` into a graph (nothing executes); ``step()`` then replays it -- or calls ``body``
    on the same stream when the body is not capturable or graphs are switched off.
Graphs are per ``attack()`` call: on this stack a replay that follows eager library work issued after the capture can
fault, so all eager work comes first and the graph is dropped at the end of the call.
"""
import warnings

import torch


class IterationGraph:
    def __init__(self, body, use_graph='auto', what='the attack iteration'):
        self.body, self.use_graph, self.what = body, use_graph, what
        self.stream = torch.cuda.Stream()
        self.graph = None
        self.reason = None

    def probe(self):
        """Two warm-up passes of ``body``; returns True when it may be captured."""
        if self.use_graph in (False, 'never'):
            self.reason = 'switched off'
            return False
        cpu_rng = torch.get_rng_state()
        prev_mode = torch.cuda.get_sync_debug_mode()
        try:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.body()  # unguarded: library handles, lazy initialisation
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")  # "prototype feature" notice
                    torch.cuda.set_sync_debug_mode("error")
                self.body()
        except Exception as e:  # noqa: BLE001
            self.reason = e
        finally:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.cuda.set_sync_debug_mode(prev_mode)
            torch.cuda.current_stream().wait_stream(self.stream)
            torch.cuda.synchronize()
            torch.set_rng_state(cpu_rng)
        if self.reason is not None and self.use_graph in (True, 'always'):
            raise RuntimeError("%s cannot be captured into a hipGraph: %r" % (self.what, self.reason))
        return self.reason is None

    def capture(self):
        """Record ``body`` (call after the state tensors have been re-initialised; nothing runs)."""
        if self.reason is not None:
            return False
        try:
            self.stream.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                self.body()
            self.graph = g
        except Exception as e:  # noqa: BLE001
            self.reason, self.graph = e, None
            torch.cuda.synchronize()
            if self.use_graph in (True, 'always'):
                raise RuntimeError("%s cannot be captured into a hipGraph: %r" % (self.what, e))
            warnings.warn("%s is not hipGraph-capturable (%r); running the eager loop" % (self.what, e))
        return self.graph is not None

    def step(self):
        """One iteration on the private stream (replay, or an eager call of ``body``)."""
        with torch.cuda.stream(self.stream):
            if self.graph is not None:
                self.graph.replay()
            else:
                self.body()

    def enter(self):
        """Make the private stream wait for work queued on the caller's stream (state initialisation)."""
        self.stream.wait_stream(torch.cuda.current_stream())

    def leave_step(self):
        """Make the caller's stream wait for the iterations queued so far (the graph stays for the next ``enter``)."""
        torch.cuda.current_stream().wait_stream(self.stream)

    def leave(self):
        """Make the caller's stream wait for the iterations; drop the graph."""
        torch.cuda.current_stream().wait_stream(self.stream)
        self.graph = None


def drive(steps):
    """Run an attack written as a generator (its stops are there for ``CW.attack_concurrently``) through to its result."""
    try:
        while True:
            next(steps)
    except StopIteration as done:
        return done.value
