"""The victim as the CW attacks call it.  A victim that offers ``attack_view()`` (PointNet: the HIP engine; DGCNN: the
folded EdgeConv view) is differentiated through that view -- the same function as the eval-mode module up to fp32
re-association, several times faster than the module's MIOpen / op-by-op formulation; its buffers are re-folded from the
module's current weights at the start of every ``attack()``.  ``fast_victim=False`` in an attack's constructor keeps the
module itself (the reference's behaviour, ShapeAttack/HiT_ADV.py has the same switch)."""


class Victim:
    def __init__(self, model, fast=True):
        self.model, self.fast, self.view = model, fast, None

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
        out = (self.view if self.view is not None else self.model)(x)
        return out[0] if isinstance(out, tuple) else out
