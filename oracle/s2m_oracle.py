"""CPU restatement of the scribble-to-mask network (DeepLabV3+ / ResNet-50) and of DAVISProcessor.to_mask.

TEST INFRASTRUCTURE ONLY (same rules as stm_oracle.py).  Functional torch over a flat state_dict; every function cites the
reference lines it restates.  PARITY PIN: oracle/make_golden.py runs the unmodified reference `model/s2m` on a seeded input and
commits the result (tests/golden/s2m_small.npz, s2m_state_dict_keys.json); tests/test_oracle_golden.py checks this file against it.
"""
import torch
import torch.nn.functional as F

from .stm_oracle import _bn, _conv, aggregate_wbg, pad_divide_by


def _bottleneck(sd, p, x, stride, dilation):
    """s2m_resnet.py:28-70: 1x1 -> 3x3 (stride, dilation, padding = dilation) -> 1x1, BN each, residual (1x1-stride + BN), ReLU."""
    y = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x)))
    y = F.relu(_bn(sd, p + "bn2", _conv(sd, p + "conv2", y, stride=stride, pad=dilation, dilation=dilation)))
    y = _bn(sd, p + "bn3", _conv(sd, p + "conv3", y))
    if (p + "downsample.0.weight") in sd:
        x = _bn(sd, p + "downsample.1", _conv(sd, p + "downsample.0", x, stride=stride))
    return F.relu(y + x)


def backbone(sd, x):
    """s2m_resnet.ResNet.forward up to layer4 with replace_stride_with_dilation=[False, False, True] (s2m_network.py:9-17,
    s2m_resnet.py:122-145: the first block of a dilated stage keeps the previous dilation) -> (layer1 out, layer4 out)."""
    p = "backbone."
    x = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x, stride=2, pad=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    low = None
    for lname, depth, stride, d_first, d_rest in (("layer1", 3, 1, 1, 1), ("layer2", 4, 2, 1, 1), ("layer3", 6, 2, 1, 1), ("layer4", 3, 1, 1, 2)):
        for b in range(depth):
            x = _bottleneck(sd, f"{p}{lname}.{b}.", x, stride if b == 0 else 1, d_first if b == 0 else d_rest)
        if lname == "layer1":
            low = x
    return low, x


def aspp(sd, x):
    """_deeplab.py:133-164 (rates 6, 12, 18; Dropout is the identity in eval)."""
    p = "classifier.aspp."
    res = [F.relu(_bn(sd, p + "convs.0.1", _conv(sd, p + "convs.0.0", x)))]
    for i, r in ((1, 6), (2, 12), (3, 18)):
        res.append(F.relu(_bn(sd, f"{p}convs.{i}.1", _conv(sd, f"{p}convs.{i}.0", x, pad=r, dilation=r))))
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(_bn(sd, p + "convs.4.2", _conv(sd, p + "convs.4.1", g)))
    res.append(F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=False))
    return F.relu(_bn(sd, p + "project.1", _conv(sd, p + "project.0", torch.cat(res, 1))))


def s2m_forward(sd, x):
    """_SimpleSegmentationModel.forward + DeepLabHeadV3Plus.forward (utils.py:15-20, _deeplab.py:45-49) -> logits [N,1,H,W]."""
    low, out = backbone(sd, x)
    low = F.relu(_bn(sd, "classifier.project.1", _conv(sd, "classifier.project.0", low)))
    a = F.interpolate(aspp(sd, out), size=low.shape[2:], mode="bilinear", align_corners=False)
    z = F.relu(_bn(sd, "classifier.classifier.1", _conv(sd, "classifier.classifier.0", torch.cat([low, a], 1), pad=1)))
    z = _conv(sd, "classifier.classifier.3", z)
    return F.interpolate(z, size=x.shape[-2:], mode="bilinear", align_corners=False)


def dilate3x3(m):
    """cv2.dilate(m, ones(3,3)) on a binary float map [...,H,W] = 3x3 max filter (borders replicate nothing: out-of-image is 0)."""
    return F.max_pool2d(m.reshape(-1, 1, *m.shape[-2:]), 3, 1, 1).reshape(m.shape)


def to_mask(sd, image, cur_mask_u8, scr_mask, k):
    """DAVISProcessor.to_mask after scribbles2mask (davis_processor.py:50-70): image [1,3,nh,nw] padded, cur_mask_u8 [1,nh,nw]
    uint8 (current argmax of the frame, padded), scr_mask int [h,w] UNPADDED rasterised scribbles (-1 = none, 0 = background
    scribble, j = scribble of object j; dilated at the true size, then padded like the reference) -> hard mask [k+1,1,nh,nw]."""
    nh, nw = image.shape[-2:]
    mask = torch.zeros((k, 1, nh, nw))
    for ki in range(1, k + 1):
        p_srb = dilate3x3((scr_mask == ki).float())
        n_srb = dilate3x3(((scr_mask != ki) & (scr_mask != -1)).float())
        rs, _ = pad_divide_by(torch.stack([p_srb, n_srb], 0)[None], 16)
        inputs = torch.cat([image, (cur_mask_u8 == ki).float()[None], rs], 1)
        mask[ki - 1] = torch.sigmoid(s2m_forward(sd, inputs))
    return aggregate_wbg(mask, keep_bg=True, hard=True)
