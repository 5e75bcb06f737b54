from . import anchor_generator, anchor_target, assigner, box_ops, coder, iou_calculator, sampler  # noqa: F401
