"""MADNet as the reference defines it (Nets/MadNet.py:73-171,173-249): layer tables, the TF variable names and the flat-buffer order of the
parameters.  Pure data + naming: no tensors, no library."""


def _r4(c):
    return (c + 3) // 4 * 4


PYR = [(3, 16, 2), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 64, 2), (64, 64, 1),
       (64, 96, 2), (96, 96, 1), (96, 128, 2), (128, 128, 1), (128, 192, 2), (192, 192, 1)]
EST = [128, 128, 96, 64, 32, 1]
CTX = [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1), (1, 1)]
LEVELS = (6, 5, 4, 3, 2)
FEAT = {6: 12, 5: 10, 4: 8, 3: 6, 2: 4}
ALPHA = 0.2   # MadNet._leaky_relu (Nets/MadNet.py:366-367)


def pyr_name(i):
    return "model/gc-read-pyramid/conv%d" % i


def est_name(k, j):
    return "model/G%d/fgc-volume-filtering-%d/disp-%d" % (k, k, j)


def ctx_name(j):
    return "model/context-%d" % j


def _merge_ranges(ranges):
    out = []
    for a, b in sorted(ranges):
        if out and a <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], b))
        else:
            out.append((a, b))
    return out


def madnet_manifest(radius_d=2, stride=1):
    """Ordered [(variable name, shape)] -- flat-buffer order.  Names are the TF variable names of
    the reference graph."""
    D = 2 * radius_d // stride + 1
    out = []

    def conv(base, k, ci, co):
        out.append((base + "/weights", (k, k, ci, co)))
        out.append((base + "/biases", (co,)))

    # the twelve pyramid layers first, then estimator by estimator (the context network behind estimator 2): the backward pass
    # finishes the estimator / context gradients BEFORE it starts on the pyramid, so [estimators | loss] is one contiguous range whose
    # all-reduce overlaps the pyramid's backward pass in the shared-model mode, and the pyramid is the other (adapter.py).  A MAD
    # block = its pyramid layers (contiguous) + its estimator (contiguous): two ranges.
    for i in range(1, 13):
        conv(pyr_name(i), 3, PYR[i - 1][0], PYR[i - 1][1])
    for k in (2, 3, 4, 5, 6):
        cin = PYR[FEAT[k] - 1][1] + D + (0 if k == 6 else 1)
        for j, co in enumerate(EST):
            conv(est_name(k, j + 1), 3, cin, co)
            cin = co
        if k == 2:
            cin = PYR[3][1] + 1
            for j, (co, _) in enumerate(CTX):
                conv(ctx_name(j + 1), 3, cin, co)
                cin = co
    return out
