"""BASELINE configs[3]: inference path -- complex_yolov4.cfg, batch 32, 608x608, model.eval()(imgs) + rotated merge-NMS
(post_processing_v2), one MI355X, fp16.  Reports images/s of the network alone and of network + NMS."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.utils.evaluation_utils import post_processing_v2_device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
torch.manual_seed(0)
model = Darknet(cfg, use_giou_loss=True, dtype='f16').cuda().eval()
model.cpu_outputs = False
x = syn.bev_images(B, 608, seed=0).cuda()
pred = syn.nms_predictions(B, 22743, 256, seed=4).cuda()        # K = 256 candidates per image


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    t_net = timeit(lambda: model(x))
    t_nms = timeit(lambda: post_processing_v2_device(pred, 0.5, 0.5))
print('inference batch %d: network %.2f ms (%.0f img/s); merge-NMS (256 candidates/img) %.2f ms; total %.0f img/s' % (
    B, 1e3 * t_net, B / t_net, 1e3 * t_nms, B / (t_net + t_nms)))
