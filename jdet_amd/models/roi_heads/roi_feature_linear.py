"""First fully connected layer on pooled RoI features, fed without a layout copy."""
from torch import nn

from jdet_amd.ops.linear import linear


class RoIFeatureLinear(nn.Linear):
    """First FC layer on pooled RoI features (R, C, PH, PW).

    The parameter is stored with its input columns in (ph, pw, c) order -- the memory order of a channels-last
    feature tensor, which is then consumed as a plain (R, PH*PW*C) matrix without a copy.  State dicts hold the
    reference's (c, ph, pw) column order (`x.flatten(1)` of an NCHW tensor, oriented_head.py:L267): the hooks below
    permute on save and load, so reference checkpoints load unchanged."""

    def __init__(self, channels, area, out_features):
        super().__init__(channels * area, out_features)
        self.channels, self.area = channels, area
        self._register_state_dict_hook(RoIFeatureLinear._to_reference_order)
        self._register_load_state_dict_pre_hook(self._from_reference_order)

    def _permute(self, w, to_reference):
        o = w.shape[0]
        if to_reference:
            return w.reshape(o, self.area, self.channels).permute(0, 2, 1).reshape(o, -1)
        return w.reshape(o, self.channels, self.area).permute(0, 2, 1).reshape(o, -1)

    @staticmethod
    def _to_reference_order(module, state_dict, prefix, local_metadata):
        key = prefix + "weight"
        if key in state_dict:
            state_dict[key] = module._permute(state_dict[key], True).contiguous()

    def _from_reference_order(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + "weight"
        if key in state_dict and tuple(state_dict[key].shape) == tuple(self.weight.shape):
            state_dict[key] = self._permute(state_dict[key], False).contiguous()

    def forward(self, x):
        if x.dim() == 4:
            # (R, C, PH, PW) -> rows in (ph, pw, c) order: a view for channels-last memory, one copy otherwise
            x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        return linear(x, self.weight, self.bias)
