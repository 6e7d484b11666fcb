#coding:utf8
"""BPR (Rendle et al.) behind the reference's plugin hooks, trained on one MI355X.

Replaces the NumPy epoch loop of the reference (recommender/cf/BPR.py:31-62).  The optional
config key ``bpr.hip`` selects how an epoch runs (existing .conf files parse unchanged):

  bpr.hip=-mode replay                  (default) negatives drawn on the host exactly as the
                                        reference draws them (random.choice + rejection,
                                        BPR.py:46-48, so a seeded `random` gives the same stream);
                                        the device applies the triplets with exact sequential
                                        semantics (yue_bpr_replay).  Same results as the reference.
  bpr.hip=-mode epoch -round auto -seed 1
                                        throughput mode: counter-based sampler on the device,
                                        S-round semantics; -round N fixes the round size
                                        (auto = the device's default: yue_default_round_events, 344,064 events on MI355X for BASELINE config 3).
  bpr.hip=-mode exact -seed 1           the device's counter-based sampler with the reference's EXACT sequential semantics (one dataflow
                                        launch per epoch, chain_kernels.hpp): the loop of BPR.py:42-58 on other negatives than
                                        Python's; needs no host sampling, so it also serves array-native data.
                                        -fast 1: the step's coefficient lr (1 - sigmoid(x)) in single precision (option chain_fast):
                                        the reference's ORDER of updates, factors within 1e-5 of its result instead of bit-equal,
                                        1.5 x the rate.
  bpr.hip=-mode adam                    the reference's LIVE path (BPR.py:65-129, a TensorFlow-1 graph): every "iteration" is one
                                        minibatch of 512 random training events x 100 rejection-sampled negatives (next_batch,
                                        :65-81, same NumPy / random calls), loss = sum softplus(-x) + regU * l2 terms, Adam(lRate)
                                        on both factor matrices (yue_adam_step), factors drawn as truncated_normal(0.005), the
                                        300-user ranking_performance after every step.  PARITY UNPINNED: TensorFlow cannot be
                                        installed here; the device path is checked against oracle/numpy_adam.py, a restatement
                                        of the graph as written (tests/test_gpu_adam.py).
  -gpu N                                HIP device ordinal.
  -topn true                            evalRanking returns a real top-N (descending, ties: lower item id)
                                        instead of the reference's order-dependent overwrite-scan.
"""
from random import choice

import numpy as np

from ...base.IterativeRecommender import IterativeRecommender
from ...data.arrays import ArrayRecord
from ...tool.config import LineConfig


def _truncated_normal(shape, stddev):
    """tf.truncated_normal: values beyond two standard deviations are drawn again (NumPy's global stream)."""
    x = np.random.normal(0.0, stddev, size=shape)
    bad = np.abs(x) > 2 * stddev
    while bad.any():
        x[bad] = np.random.normal(0.0, stddev, size=int(bad.sum()))
        bad = np.abs(x) > 2 * stddev
    return x.astype(np.float32)


class BPR(IterativeRecommender):

    def __init__(self, conf, trainingSet=None, testSet=None, fold='[1]'):
        super(BPR, self).__init__(conf, trainingSet, testSet, fold)

    def initModel(self):
        super(BPR, self).initModel()
        self.m = self.data.getSize('user')
        self.n = self.data.getSize(self.recType)
        self.train_size = len(self.data.trainingData)
        if self._options()['-mode'] == 'adam':               # BPR.py:97-98: tf.truncated_normal(stddev=0.005), U first
            self.P = _truncated_normal((self.m, self.k), 0.005)
            self.Q = _truncated_normal((self.n, self.k), 0.005)

    def _options(self):
        opts = {'-mode': 'replay', '-round': 'auto', '-seed': '1', '-fast': '0'}
        if self.config.contains('bpr.hip'):
            given = LineConfig(self.config['bpr.hip'])
            for key in opts:
                if given.contains(key):
                    opts[key] = given[key]
        return opts

    def _draw_negatives(self, itemList, listened_names):
        """BPR.py:42-49: one rejection-sampled negative per training event, reference order."""
        item_ids = self.data.name2id[self.recType]
        out = []
        for user in self.data.userRecord:
            mine = listened_names[user]
            for _ in self.data.userRecord[user]:
                item_j = choice(itemList)
                while item_j in mine:
                    item_j = choice(itemList)
                out.append(item_ids[item_j])
        return np.asarray(out, np.int32)

    def _next_batch(self, listened):
        """BPR.py:65-81: 512 random training events, 100 rejection-sampled negatives each (random.randint over item ids)."""
        from random import randint
        train = self.data.trainingData
        batch_idx = np.random.randint(len(train), size=512)
        user_idx, item_idx, neg_idx = [], [], []
        for idx in batch_idx:
            uid = self.data.getId(train[idx]['user'], 'user')
            tid = self.data.getId(train[idx][self.recType], self.recType)
            mine = listened[uid]
            for _ in range(100):
                item_j = randint(0, self.n - 1)
                while item_j in mine:
                    item_j = randint(0, self.n - 1)
                user_idx.append(uid)
                item_idx.append(tid)
                neg_idx.append(item_j)
        return user_idx, item_idx, neg_idx

    def _build_adam(self):
        """The live path of the reference (BPR.py:83-129) with the graph evaluated by yue_adam_step."""
        listened = {}
        for user in self.data.userRecord:                        # :84-91
            uid = self.data.getId(user, 'user')
            listened[uid] = {self.data.getId(ev[self.recType], self.recType) for ev in self.data.userRecord[user]}
        self._sync_factors_to_device()
        dev = self.dev
        dev.adam_reset()
        for epoch in range(self.maxIter):
            user_idx, item_idx, neg_idx = self._next_batch(listened)
            loss = dev.adam_step(user_idx, item_idx, neg_idx, self.lRate, self.regU, epoch + 1)
            print('iteration:', epoch, 'loss:', loss)
            self.loss = loss
            self.ranking_performance()                           # :129, the 300-user check after every step
        dev.get_factors(self.P, self.Q)
        self._device_factors_current = True

    def buildModel(self):
        opts = self._options()
        if opts['-mode'] == 'adam':
            if isinstance(self.data, ArrayRecord):
                print('bpr.hip=-mode adam samples from the text log\'s training events; array-native data needs -mode epoch')
                exit(-1)
            return self._build_adam()
        if isinstance(self.data, ArrayRecord) and opts['-mode'] == 'replay':
            print('array-native data needs bpr.hip=-mode epoch or exact (the replay mode samples over item names)')
            exit(-1)
        if opts['-mode'] not in ('replay', 'epoch', 'exact'):
            print('bpr.hip: unknown -mode ' + opts['-mode'])
            exit(-1)
        self._sync_factors_to_device()
        dev, arr = self.dev, self._arrays
        dev.set_option('epoch_exact', 1 if opts['-mode'] == 'exact' else 0)
        dev.set_option('chain_fast', 1 if opts['-fast'] not in ('0', '', 'off', 'false') else 0)
        if opts['-mode'] == 'replay':
            ev_u = np.repeat(np.arange(self.m, dtype=np.int32), np.diff(arr['ev_ptr']))
            listened_names = {user: {ev[self.recType] for ev in events} for user, events in self.data.userRecord.items()}
            itemList = list(self.data.name2id[self.recType].keys())
        print('training...')
        iteration = 0
        while iteration < self.maxIter:
            if opts['-mode'] == 'replay':
                j = self._draw_negatives(itemList, listened_names)
                nll = dev.bpr_replay(ev_u, arr['ev_i'], j, self.lRate, self.regU, self.regI)
                sumP, sumQ = dev.sumsq()
            else:
                nll, sumP, sumQ = dev.bpr_epoch(int(opts['-seed']), iteration, 0 if opts['-round'] == 'auto' else int(opts['-round']), self.lRate, self.regU, self.regI)
            # BPR.py:58-59.  Under NumPy 2 `regU * (P*P).sum()` is a float32 and turns the loss into
            # a float32; the same promotion is applied here so the printed line matches.
            self.loss = nll + (self.regU * np.float32(sumP) + self.regI * np.float32(sumQ))
            iteration += 1
            if self.isConverged(iteration):
                break
        dev.get_factors(self.P, self.Q)          # state contract: trained factors back on the host
        self._device_factors_current = True

    def predict(self, u):
        'invoked to rank all the items for the user'
        return super(BPR, self).predict(u)
