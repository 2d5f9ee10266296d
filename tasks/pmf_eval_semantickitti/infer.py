"""Point-wise inference on SemanticKITTI (counterpart of the reference's tasks/pmf_eval_semantickitti/infer.py:16-330).

Per frame (:67-146): perspective projection (return_uproj) -> zero padding by (h_pad, w_pad) -> normalisation * mask ->
PMFNet (HIP plan, eval) -> crop back -> argmax -> labels of the original points, either read at their pixel or voted by
the KNN post-processing (pc_processor/postproc/knn.py) -> learning ids to annotation ids (learning_map_inv) -> int32
``<save_path>/preds/sequences/<seq>/predictions/<frame>.label``; point-wise and pixel-wise confusion matrices when
labels exist (:146-330: mean / per-class IoU, accuracy, recall, class distribution, frequency-weighted IoU).

    python infer.py config_server_kitti.yaml
"""
import argparse
import datetime
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import pc_processor  # noqa: E402
from option import Option  # noqa: E402


def _table(header, rows):
    w = [max(len(str(x)) for x in col) for col in zip(header, *rows)] if rows else [len(h) for h in header]
    line = lambda r: " | ".join(str(x).ljust(n) for x, n in zip(r, w))
    return "\n".join([line(header), "-+-".join("-" * n for n in w)] + [line(r) for r in rows])


class Inference(object):
    def __init__(self, settings, model, recorder):
        self.settings, self.recorder = settings, recorder
        self.model = model.cuda()
        self.knn_flag = settings.config["post"]["KNN"]["use"]
        self.knn_post = pc_processor.postproc.KNN(params=settings.config["post"]["KNN"]["params"],
                                                  nclasses=settings.n_classes)
        self.val_loader, self.salsa_loader = self._initDataloader()
        self.prediction_path = os.path.join(settings.save_path, "preds")
        self.evaluator = pc_processor.metrics.IOUEval(n_classes=settings.n_classes, device=torch.device("cpu"), ignore=[0])
        self.pixel_eval = pc_processor.metrics.IOUEval(n_classes=settings.n_classes, device=torch.device("cpu"), ignore=[0])
        if self.knn_flag:
            self.recorder.logger.info("using KNN Post Process")

    def _initDataloader(self):
        s = self.settings
        if s.dataset != "SemanticKitti":
            raise ValueError("invalid dataset: {}".format(s.dataset))
        cfg_path = s.config.get("data_config_path") or pc_processor.dataset.semantic_kitti.DEFAULT_CONFIG
        valset = pc_processor.dataset.semantic_kitti.SemanticKitti(
            root=s.data_root, sequences=list(s.config.get("sequences", {}).get("valid", [8])), config_path=cfg_path,
            has_label=s.has_label, has_image=True)
        loader = pc_processor.dataset.PerspectiveViewLoader(dataset=valset, config=s.config, is_train=False,
                                                            return_uproj=True)
        return loader, loader            # frames are visited in order, one at a time, on the device (:58-63: batch 1)

    @torch.no_grad()
    def run(self):
        s = self.settings
        self.model.eval()
        self.evaluator.reset()
        self.pixel_eval.reset()
        sensor = s.config["sensor"]
        mean = torch.tensor(sensor["img_mean"], dtype=torch.float32).view(1, -1, 1, 1).cuda()
        std = torch.tensor(sensor["img_stds"], dtype=torch.float32).view(1, -1, 1, 1).cuda()
        h_pad, w_pad = sensor["h_pad"], sensor["w_pad"]
        ds = self.salsa_loader.dataset
        t_start = time.time()
        for i in range(len(self.val_loader)):
            feat, mask, label, ux, uy, udepth = self.val_loader[i]
            t0 = time.time()
            ux, uy = ux.long(), uy.long()
            if udepth.shape[0] != ux.shape[0]:            # uncropped sweep: depth of the projected points only
                udepth = udepth[self.salsa_loader.last_keep]
            proj_depth = feat[0].clone()
            proj_depth = proj_depth - proj_depth.eq(0).float()                    # -1 on empty pixels (:78-79)
            pad = (w_pad, w_pad, h_pad, h_pad)
            x = torch.nn.functional.pad(feat[None], pad)
            m = torch.nn.functional.pad(mask[None], pad)
            x[:, 0:5] = (x[:, 0:5] - mean) / std * m.unsqueeze(1)
            pred, _ = self.model(x[:, 0:5], x[:, 5:8])
            pred = pred[:, :, h_pad:h_pad + label.size(0), w_pad:w_pad + label.size(1)]
            if self.knn_flag and not s.has_label and pred.is_contiguous() and hasattr(self.knn_post, "batch_prob"):
                # no pixel-wise evaluation wanted: class argmax (int32 map) + vote inside the library, no torch.argmax
                unproj = self.knn_post.batch_prob(pred, [(proj_depth, udepth, uy, ux)])[0]
            else:
                argmax = pred.argmax(dim=1)
                if s.has_label:
                    self.pixel_eval.addBatch(argmax, label.long()[None])
                if self.knn_flag:
                    unproj = self.knn_post(proj_depth, udepth, argmax[0], uy, ux)    # (:104-110: x index = column)
                else:
                    unproj = argmax[0][ux, uy]
            pred_np = unproj.cpu().numpy().reshape(-1).astype(np.int32)
            if s.has_label:
                # predictions cover the points that project into the image, in file order (keep mask of the loader)
                sem, _ = ds.loadLabelByIndex(i)
                kept = self._kept_labels(i, sem, pred_np.shape[0])
                self.evaluator.addBatch(pred_np, ds.class_map_lut[kept])
            path = pc_processor.dataset.semantic_kitti.write_prediction(ds, i, pred_np, self.prediction_path)
            msg = "Iter [{:04d}|{:04d}] Datatime: {:0.3f} ProcessTime: {:0.3f} -> {}".format(
                i, len(self.val_loader), t0 - t_start, time.time() - t0, os.path.basename(path))
            if s.has_label:
                msg += " meanIOU {:0.4f}".format(self.pixel_eval.getIoU()[0].item())
            print(msg)
            t_start = time.time()
            if s.is_debug:
                break
        if s.has_label:
            self.report("Point-wise Evaluation Results (3D eval)", self.evaluator)
            self.report("Pixel-wise Evaluation Results (2D eval)", self.pixel_eval)

    def _kept_labels(self, index, sem, n_pred):
        """labels of the points the loader kept (x > 0.5 and inside the image), in file order.  The reference's FOV
        dataset (tasks/process_semantickitti_fov) stores only such points, so its files line up one to one (:118-122);
        for uncropped sweeps the loader's keep mask selects them."""
        if sem.shape[0] == n_pred:
            return sem
        keep = self.salsa_loader.last_keep
        return sem[keep.cpu().numpy()]

    def report(self, title, ev):
        log = self.recorder.logger.info
        names = self.salsa_loader.dataset.mapped_cls_name
        n = self.settings.n_classes
        m_acc, c_acc = ev.getAcc()
        m_rec, c_rec = ev.getRecall()
        m_iou, c_iou = ev.getIoU()
        log("============== {} ===================".format(title))
        log("Acc avg: {:.4f}, IOU avg: {:.4f}, Recall avg: {:.4f}".format(m_acc.item(), m_iou.item(), m_rec.item()))
        log("\n" + _table(["ClassIdx", "class_name", "IOU", "Acc", "Recall"],
                          [[i, names[i], "%.4f" % c_iou[i].item(), "%.4f" % c_acc[i].item(), "%.4f" % c_rec[i].item()]
                           for i in range(1, n)]))
        log("latex: " + "".join(" & {:0.1f}".format(c_iou[i].item() * 100) for i in range(1, n)) +
            " & {:0.1f}".format(m_iou.item() * 100))
        conf = ev.conf_matrix.clone().cpu()
        conf[0] = 0
        conf[:, 0] = 0
        dist = conf.sum(0).double()
        log("\n" + _table(["Class Name", "Number of points", "Percentage"],
                          [[names[i], int(dist[i].item()), "%.4f" % (dist[i] / dist.sum().clamp_min(1)).item()]
                           for i in range(n)]))
        freqw = dist[1:] / dist[1:].sum().clamp_min(1)
        log("fwIoU: {}".format((c_iou[1:].cpu().double() * freqw).sum().item()))


class Experiment(object):
    def __init__(self, settings):
        self.settings = settings
        settings.check_path()
        torch.manual_seed(settings.seed)
        torch.cuda.manual_seed(settings.seed)
        torch.cuda.set_device(0)
        self.recorder = pc_processor.checkpoint.Recorder(settings, settings.save_path, use_tensorboard=False)
        self.model = pc_processor.models.PMFNet(
            pcd_channels=5, img_channels=3, nclasses=settings.n_classes, base_channels=settings.base_channels,
            image_backbone=settings.img_backbone, imagenet_pretrained=settings.imagenet_pretrained)
        if settings.pretrained_model is not None:
            if not os.path.isfile(settings.pretrained_model):
                raise FileNotFoundError("pretrained model not found: {}".format(settings.pretrained_model))
            self.model.load_state_dict(torch.load(settings.pretrained_model, map_location="cpu"))
            self.recorder.logger.info("loading pretrained weight from: {}".format(settings.pretrained_model))
        self.inference = Inference(settings, self.model, self.recorder)

    def run(self):
        t0 = time.time()
        self.inference.run()
        self.recorder.logger.info("==== total cost time: {}".format(datetime.timedelta(seconds=time.time() - t0)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="PMF inference on MI355X")
    ap.add_argument("config_path", type=str, metavar="config_path")
    ap.add_argument("--id", type=int, default=0)
    exp = Experiment(Option(ap.parse_args().config_path))
    print("===init env success===")
    exp.run()
