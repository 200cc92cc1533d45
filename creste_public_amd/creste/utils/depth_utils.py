"""Depth discretisation helpers -- mirrors the torch part of
/root/reference/creste/utils/depth_utils.py: convert_to_metric_depth_differentiable (:300-313),
convert_to_metric_depth (:316-343), bin_depths (:346-383).  On the HIP path the softmax expectation
is the `creste_depth_expectation_f32` kernel; these host versions serve label preparation."""
import math

import torch
import torch.nn.functional as F


def convert_to_metric_depth_differentiable(depth_logits, mode, depth_min, depth_max, num_bins):
    probs = F.softmax(depth_logits, dim=1)
    vals = torch.linspace(depth_min, depth_max, num_bins, device=depth_logits.device).view(1, -1, 1, 1)
    return torch.sum(probs * vals, dim=1)


def convert_to_metric_depth(depth_bin, mode, depth_min, depth_max, num_bins):
    if mode == "UD":
        return depth_bin * ((depth_max - depth_min) / num_bins) + depth_min
    if mode == "LID":
        bs = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        return depth_min + 0.5 * bs * depth_bin * (depth_bin + 1)
    if mode == "SID":
        return (math.exp(math.log(1 + depth_max) - math.log(1 + depth_min)) * depth_bin / num_bins) + \
            math.log(1 + depth_min)
    raise NotImplementedError


def bin_depths(depth_map, mode, depth_min, depth_max, num_bins, target=False):
    if mode == "UD":
        idx = (depth_map - depth_min) / ((depth_max - depth_min) / num_bins)
    elif mode == "LID":
        bs = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        idx = -0.5 + 0.5 * torch.sqrt(1 + 8 * (depth_map - depth_min) / bs)
    elif mode == "SID":
        idx = num_bins * (torch.log(1 + depth_map) - math.log(1 + depth_min)) / \
            (math.log(1 + depth_max) - math.log(1 + depth_min))
    else:
        raise NotImplementedError
    if target:
        bad = (idx < 0) | (idx > num_bins) | (~torch.isfinite(idx))
        idx[bad] = num_bins
        idx = idx.type(torch.int64)
    return idx
