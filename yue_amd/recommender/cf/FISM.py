#coding:utf8
"""FISM (Kabbur et al., factored item similarity) behind the reference's plugin hooks -- parity path.

Replaces the NumPy loop of the reference's recommender/cf/FISM.py:27-73 and its predict (:75-83) with
the device functions yue_fism_* (include/yue_hip.h).  Model arrays keep the reference's types: ``P``
float64 [items, k] (item-history factors), ``Q`` float32 [items, k], ``Bi`` float64 [items]
(FISM.py:15-18).  Negatives are drawn on the host exactly as the reference draws them (``choice`` over
the item names + rejection, FISM.py:50-53), so a seeded ``random`` gives the reference's stream; the
device then runs the epoch in the reference's strictly sequential order.  Config: the reference's
``FISM=-rho R -alpha A`` line; ``bpr.hip=-gpu N`` selects the device as for BPR.  One optional addition:
``FISM=-rho R -alpha A -round U`` trains in rounds of U users (throughput form, yue_fism_rounds:
every user of a round starts from the round-start model; U = 1, the default, is the reference's pass).
"""
from random import choice

import numpy as np

from ...base.IterativeRecommender import IterativeRecommender
from ...tool.config import LineConfig


class FISM(IterativeRecommender):

    def __init__(self, conf, trainingSet=None, testSet=None, fold='[1]'):
        super(FISM, self).__init__(conf, trainingSet, testSet, fold)

    def readConfiguration(self):
        super(FISM, self).readConfiguration()
        options = LineConfig(self.config['FISM'])
        self.rho = max(1, int(options['-rho']))
        self.alpha = float(options['-alpha'])
        self.roundUsers = max(1, int(options['-round'])) if options.contains('-round') else 1

    def initModel(self):
        super(FISM, self).initModel()                    # draws the base class's P, Q first (kept: same random stream)
        n = self.data.getSize(self.recType)
        self.Bi = np.random.rand(n) / 100
        self.P = np.random.rand(n, self.k) / 100
        self._fism_current = False

    # ---- device state ---------------------------------------------------------------------
    def _events(self):
        """(user_ptr, ev_i) in the reference's processing order (users as in userRecord, events as recorded)."""
        if not hasattr(self, '_fism_events'):
            arrays = self.data.to_arrays(self.recType)
            self._fism_events = (arrays['ev_ptr'], arrays['ev_i'])
        return self._fism_events

    def _upload(self):
        if not getattr(self, '_fism_current', False):
            self._device().fism_set_model(self.P, self.Q, self.Bi)
            self._fism_current = True

    def _draw_negatives(self, itemList, listened):
        """FISM.py:39-53: rho rejection-sampled negatives per event, users with one event skipped."""
        item_ids = self.data.name2id[self.recType]
        out = []
        for user in self.data.userRecord:
            events = self.data.userRecord[user]
            if len(events) == 1:
                continue
            mine = listened[user]
            for _ in events:
                for _count in range(self.rho):
                    item_j = choice(itemList)
                    while item_j in mine:
                        item_j = choice(itemList)
                    out.append(item_ids[item_j])
        return np.asarray(out, np.int32)

    def buildModel(self):
        user_ptr, ev_i = self._events()
        sizes = np.diff(user_ptr)
        coef = np.array([pow(int(nu) - 1, -self.alpha) if nu > 1 else 0.0 for nu in sizes], np.float64)    # FISM.py:42
        listened = {user: {ev[self.recType] for ev in events} for user, events in self.data.userRecord.items()}
        itemList = list(self.data.name2id[self.recType].keys())
        self._upload()
        dev = self.dev
        print('training...')
        iteration = 0
        while iteration < self.maxIter:
            negs = self._draw_negatives(itemList, listened)
            if self.roundUsers > 1:      # throughput form: rounds of users
                half_sq, sumP, sumQ, sumB = dev.fism_rounds(user_ptr, ev_i, negs, self.rho, coef, self.roundUsers, self.lRate, self.regI, self.regB)
            else:                        # the reference's strictly sequential pass
                half_sq, sumP, sumQ, sumB = dev.fism_epoch(user_ptr, ev_i, negs, self.rho, coef, self.lRate, self.regI, self.regB)
            # FISM.py:70 with NumPy's scalar types: (P*P).sum() and Bi.dot(Bi) are float64, (Q*Q).sum() is a float32
            # (so `regI * ...` is a float32); the total stays float64
            self.loss = np.float64(half_sq) + (self.regU * np.float64(sumP) + self.regI * np.float32(sumQ) + self.regB * np.float64(sumB))
            iteration += 1
            if self.isConverged(iteration):
                break
        dev.fism_get_model(self.P, self.Q, self.Bi)      # state contract: trained arrays back on the host

    # ---- scoring --------------------------------------------------------------------------
    def _rows_of(self, users):
        user_ptr, ev_i = self._events()
        uids = [self.data.getId(u, 'user') for u in users]
        rows = [ev_i[user_ptr[u]:user_ptr[u + 1]] for u in uids]
        ptr = np.zeros(len(rows) + 1, np.int64)
        ptr[1:] = np.cumsum([len(r) for r in rows])
        return ptr, (np.concatenate(rows) if rows else np.zeros(0, np.int32))

    def predict(self, user):
        'scores of all items for one user, item-id order (float64)'
        self._upload()
        _, items = self._rows_of([user])
        return self.dev.fism_scores(items)

    def _scan(self, users, N, mask=None):
        if mask is not None:
            print('FISM: ranking_performance (test items masked) is not part of this build')
            exit(-1)
        self._upload()
        ptr, items = self._rows_of(users)
        out = np.zeros((len(users), N), np.int32)
        step = max(1, (1 << 30) // max(1, self.data.getSize(self.recType)))      # users x items per call below 2^31
        for a in range(0, len(users), step):
            b = min(len(users), a + step)
            out[a:b] = self.dev.fism_topn_scan(ptr[a:b + 1] - ptr[a], items[ptr[a]:ptr[b]], N)[0]
        return out
