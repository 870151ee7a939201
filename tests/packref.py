"""numpy mirror of the device kernel k_pack_records (ffgpu_pack_records) -- test infrastructure: the product packs on the
GPU only; the host side of the product (ffcnn_amd/dist.py) only UNPACKS what was gathered."""
import numpy as np

from ffcnn_amd.dist import packed_bytes


def pack_records(recs, cap):
    """numpy mirror of the device kernel: recs = structured array (count, ncand, overflow, reserved, box[MAX_DET]) of one step."""
    batch = len(recs)
    box_dtype = recs.dtype["box"].subdtype[0]
    out = np.zeros(packed_bytes(batch, cap, box_dtype.itemsize), np.uint8)
    hdr = out[:16].view(np.int32)
    fr = out[16:16 + 16 * batch].view(np.int32).reshape(batch, 4)
    box = out[16 + 16 * batch:16 + 16 * batch + box_dtype.itemsize * cap].view(box_dtype)
    first = 0
    for n in range(batch):
        cnt = int(recs[n]["count"])
        kept = max(0, min(cnt, cap - first))
        fr[n] = (kept, recs[n]["ncand"], int(recs[n]["overflow"]) | (2 if kept < cnt else 0), int(recs[n]["nfull"]))
        box[min(first, cap):min(first, cap) + kept] = recs[n]["box"][:kept]
        first += cnt
    hdr[:] = (min(first, cap), int(first > cap), batch, cap)
    return out
