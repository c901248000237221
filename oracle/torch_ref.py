"""Second CPU implementation of the same graph on torch-CPU ops (oneDNN), NCHW.

Test infrastructure only (see ``oracle/__init__.py``).  Purpose: (a) an implementation
of the conv stack that shares no arithmetic code with ``oracle/nn.py`` (cross-check of the
restated TF semantics: SAME padding, BN, pooling, nearest upsample), and (b) the multi-core
CPU baseline timed next to the GPU numbers (torch-CPU/oneDNN is the closest
available stand-in for the reference's TF-CPU kernels).  Follows the same reference lines as
``oracle/model.py`` (code/yolo3/model.py:14-30,91-168,170-342; efficientnet.py:406-536,611-710).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .model import (BN_EPS, EFFNET_COEFFS, EFFNET_STAGES, MBV2_BLOCKS, _make_divisible, round_filters,
                    round_repeats)


class _Net:
    def __init__(self, P, dtype=torch.float32):
        self.P, self.dtype, self.cache = P, dtype, {}

    def _t(self, key, fn):
        v = self.cache.get(key)
        if v is None:
            v = self.cache[key] = fn()
        return v

    @staticmethod
    def same_pad(x, k, s):
        h, w = x.shape[2:]
        def amt(n):
            out = -(-n // s)
            tot = max((out - 1) * s + k - n, 0)
            return tot // 2, tot - tot // 2
        (pt, pb), (pl, pr) = amt(h), amt(w)
        return F.pad(x, (pl, pr, pt, pb)) if (pt or pb or pl or pr) else x

    def conv(self, name, x, cout, k=1, s=1, bias=False):
        cin = x.shape[1]
        w = self._t(name + '/k', lambda: torch.from_numpy(self.P.conv(name, k, cin, cout)).permute(3, 2, 0, 1)
                    .contiguous().to(self.dtype))
        b = self._t(name + '/b', lambda: torch.from_numpy(self.P.bias(name, cout)).to(self.dtype)) if bias else None
        if k > 1:
            x = self.same_pad(x, k, s)
        return F.conv2d(x, w, b, stride=s)

    def dw(self, name, x, k, s):
        c = x.shape[1]
        w = self._t(name + '/dk', lambda: torch.from_numpy(self.P.dw(name, k, c)).permute(2, 0, 1).unsqueeze(1)
                    .contiguous().to(self.dtype))
        return F.conv2d(self.same_pad(x, k, s), w, None, stride=s, groups=c)

    def bn(self, name, x):
        c = x.shape[1]
        g, b, m, v = self._t(name + '/bn', lambda: [torch.from_numpy(a).to(self.dtype) for a in self.P.bn(name, c)])
        return F.batch_norm(x, m, v, g, b, False, 0.0, BN_EPS)

    # ---- blocks
    def mbv2(self, x, alpha):
        acts = {}
        x = F.relu6(self.bn('bn_Conv1', self.conv('Conv1', x, _make_divisible(32 * alpha, 8), 3, 2)))
        for b, (f, s, t) in enumerate(MBV2_BLOCKS[:16]):
            p = 'expanded_conv_' if b == 0 else 'block_%d_' % b
            cin, cout, inp = x.shape[1], _make_divisible(int(f * alpha), 8), x
            if b > 0:
                x = F.relu6(self.bn(p + 'expand_BN', self.conv(p + 'expand', x, t * cin)))
            x = F.relu6(self.bn(p + 'depthwise_BN', self.dw(p + 'depthwise', x, 3, s)))
            x = self.bn(p + 'project_BN', self.conv(p + 'project', x, cout))
            if cin == cout and s == 1:
                x = inp + x
            acts[b] = x
        return acts[15], acts[12], acts[5], F.max_pool2d(acts[2], 4)

    def mbconv(self, name, x, k, s, e, cin, cout, se, lite=False):
        act = F.relu6 if lite else F.silu
        inp = x
        if e != 1:
            x = act(self.bn(name + '_expand_BN', self.conv(name + '_expand', x, cin * e)))
        x = act(self.bn(name + '_dw_BN', self.dw(name + '_dw', x, k, s)))
        if se and not lite:
            r = max(1, int(cin * se))
            g = x.mean((2, 3), keepdim=True)
            g = F.silu(self.conv(name + '_se_reduce', g, r, bias=True))
            g = torch.sigmoid(self.conv(name + '_se_expand', g, x.shape[1], bias=True))
            x = g * x
        x = self.bn(name + '_project_BN', self.conv(name + '_project', x, cout))
        if s == 1 and cin == cout:
            x = x + inp
        return x

    def effnet(self, x, width, depth, lite):
        act = F.relu6 if lite else F.silu
        x = act(self.bn('stem_BN', self.conv('stem_conv', x, round_filters(32, width), 3, 2)))
        ends = {}
        for si, (r, k, s, e, i, o, se) in enumerate(EFFNET_STAGES[:6], start=1):
            i, o, r = round_filters(i, width), round_filters(o, width), round_repeats(r, depth)
            x = self.mbconv('stage%d_block0' % si, x, k, s, e, i, o, se, lite)
            for rep in range(1, r):
                x = self.mbconv('stage%d_block%d' % (si, rep), x, k, 1, e, o, o, se, lite)
            ends[si] = x
        return ends[6], ends[5], ends[3], F.max_pool2d(ends[2], 4)

    def head_block(self, name, x, f, out_f):
        x = F.relu6(self.bn(name + '_conv_BN', self.conv(name + '_conv', x, f)))
        x = self.mbconv(name + '_mb', x, 3, 1, 1, f, out_f, 0.25)
        return x, self.conv(name + '_y', x, out_f)

    def cbr(self, name, bnname, x, f):
        return F.relu6(self.bn(bnname, self.conv(name, x, f)))

    def body(self, x, model_name, num_anchors, num_classes):
        out_f = num_anchors * (num_classes + 5)
        if model_name.startswith('mobilenetv2'):
            b1, b2, b3, b4 = self.mbv2(x, 0.75 if model_name == 'mobilenetv2x75' else 1.4)
        else:
            name, lite = (model_name[:-5], True) if model_name.endswith('-lite') else (model_name, False)
            w, d = EFFNET_COEFFS['efficientnet-b' + name[-1]]
            b1, b2, b3, b4 = self.effnet(x, w, d, lite)
        up = lambda t: F.interpolate(t, scale_factor=2, mode='nearest')
        c = [self.conv('rfcr_b%dc' % (i + 1), t, 48) for i, t in enumerate((b1, b2, b3, b4))]
        a = self.P.alpha('rfcr_wsum')
        bc = float(a[0]) * up(c[0]) + float(a[1]) * c[1] + float(a[2]) * F.max_pool2d(c[2], 2) + float(a[3]) * c[3]
        bc = F.relu6(self.bn('rfcr_sep_dw_BN', self.dw('rfcr_sep_dw', bc, 5, 1)))
        bc = F.relu6(self.bn('rfcr_sep_pw_BN', self.conv('rfcr_sep_pw', bc, 96)))
        b1 = torch.cat([b1, F.max_pool2d(bc, 2)], 1)
        b2 = torch.cat([b2, bc], 1)
        b3 = torch.cat([b3, up(bc)], 1)
        x, _ = self.head_block('td1', b1, 512, out_f); c1 = x
        x = self.cbr('block_20_conv', 'block_20_BN', x, 256)
        x, _ = self.head_block('td2', torch.cat([up(x), b2], 1), 256, out_f); c2 = x
        x = self.cbr('block_24_conv', 'block_24_BN', x, 128)
        x, _ = self.head_block('td3', torch.cat([up(x), b3], 1), 128, out_f); c3 = x
        x, y3 = self.head_block('bu3', c3, 128, out_f)
        x = self.cbr('bu3_down_conv', 'bu3_down_BN', x, 128)
        x, y2 = self.head_block('bu2', torch.cat([F.max_pool2d(x, 2), c2], 1), 256, out_f)
        x = self.cbr('bu2_down_conv', 'bu2_down_BN', x, 256)
        x, y1 = self.head_block('bu1', torch.cat([F.max_pool2d(x, 2), c1], 1), 512, out_f)
        return [y1, y2, y3]


class TorchReference:
    """Callable: NHWC float32 numpy images -> [y1,y2,y3] numpy [B,G,G,A,C+5]."""

    def __init__(self, P, model_name, num_anchors=3, num_classes=20, dtype=torch.float32):
        self.net = _Net(P, dtype)
        self.model_name, self.a, self.c = model_name, num_anchors, num_classes

    @torch.no_grad()
    def __call__(self, x_nhwc):
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).to(self.net.dtype).permute(0, 3, 1, 2)
        ys = self.net.body(x, self.model_name, self.a, self.c)
        out = []
        for y in ys:
            y = y.permute(0, 2, 3, 1).contiguous()
            out.append(y.reshape(y.shape[0], y.shape[1], y.shape[2], self.a, self.c + 5).numpy())
        return out
