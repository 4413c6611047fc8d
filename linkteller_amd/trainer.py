"""The slice of ``GCNTrainer`` the attack goes through (reference gcn_trainer.py:55-110, 113-141,
262-284, 320-406): build the model, load a reference-trained ``state_dict``, one clean forward for
the utility metrics, dispatch to ``Attacker``.  Training is outside the hot path."""
from __future__ import annotations

import logging
import time

import torch
import torch.nn.functional as F
from sklearn.metrics import average_precision_score

from .attacker import Attacker
from .gcn import GCN, GCN3


class GCNTrainer:
    def __init__(self, args, subdir="", worker=None):
        self.args = args
        self.worker = worker
        self.mode = worker.mode
        self.dataset = worker.dataset
        self.subdir = subdir

    def init_model(self, model_path=""):
        a, w = self.args, self.worker
        if self.mode not in ("vanilla-clean", "vanilla"):
            raise NotImplementedError("mode = {} no corrsponding model!".format(self.mode))
        if a.n_layer == 2:
            self.model = GCN(nfeat=w.n_features, nhid=a.hidden, nclass=w.n_classes, dropout=a.dropout)
        elif a.n_layer == 3:
            self.model = GCN3(nfeat=w.n_features, nhid1=a.hidden1, nhid2=a.hidden2, nclass=w.n_classes,
                              dropout=a.dropout)
        else:
            raise NotImplementedError(f"n_layer = {a.n_layer} not implemented!")
        if not model_path:
            raise NotImplementedError("training is out of scope: pass --model-path to a state_dict trained "
                                      "with the reference (gcn_trainer.py:240)")
        self.model.load_state_dict(torch.load(model_path, map_location="cpu"))
        print("load model from {} done!".format(model_path))
        self.model_path = model_path
        if torch.cuda.is_available():
            self.model.cuda()

    def forward(self, mode="train"):
        w = self.worker
        if not w.transfer:
            raise NotImplementedError(f"dataset = {self.dataset} not implemented!")
        return self.model(w.features_1, w.adj_1) if mode == "train" else self.model(w.features_2, w.adj_2)

    def rare_class_f1(self, output, labels):
        """gcn_trainer.py:262-284: F1 / precision / recall / AP of the minority class."""
        ind = [torch.where(labels == 0)[0], torch.where(labels == 1)[0]]
        rare = int(len(ind[0]) > len(ind[1]))
        conf, pred = F.softmax(output, dim=1).max(1)
        ap = average_precision_score(labels.cpu() if rare == 1 else 1 - labels.cpu(), conf.detach().cpu())
        pred = pred.type_as(labels)
        tp = torch.sum(pred[ind[rare]] == rare).item()
        t, p = len(ind[rare]), torch.sum(pred == rare).item()
        if p == 0:
            return 0
        precision, recall = tp / p, tp / t
        return 2 * precision * recall / (precision + recall), precision, recall, ap

    def eval_output(self, output, mode="clean", eval_degree=False):
        a = self.args
        if a.attack:
            self.attacker = Attacker(args=a, model=self.model, worker=self.worker)
            self.attacker.prepare_test_data()
            t = time.time()
            if a.attack_mode == "efficient":                      # gcn_trainer.py:326-337
                if a.sample_type == "balanced-full":
                    self.attacker.link_prediction_attack_efficient_balanced()
                else:
                    self.attacker.link_prediction_attack_efficient()
            elif a.attack_mode in ("baseline", "baseline-feat"):
                if a.sample_type == "balanced-full":
                    self.attacker.baseline_attack_balanced()
                else:
                    self.attacker.baseline_attack()
            else:
                raise NotImplementedError(f"attack_mode = {a.attack_mode}: the per-pair naive attack is superseded by "
                                          "the efficient one (SURVEY.md section 2)")
            print(f"attacks done using {time.time() - t} seconds!")
        labels = self.worker.labels_2
        loss_test = F.cross_entropy(output, labels.squeeze())
        acc = self.rare_class_f1(output, labels)
        info = f"[{mode}] Test set results: loss = {loss_test.item():.4f} "
        if acc != 0:
            info += f"rare_class_f1 = {acc[0]:.4f} prec = {acc[1]:.4f} reca = {acc[2]:.4f} ap_score = {acc[3]:.4f}"
        print(info)
        logging.info(info)

    def test(self, eval_degree=False):
        self.model.eval()
        with torch.no_grad():
            output = self.forward(mode="test")
        self.eval_output(output, "clean", eval_degree=eval_degree)
