"""Matrix-factorisation base class: hyper-parameters, P/Q init, bold-driver learning rate,
convergence test, and the ranking evaluation -- the surface of the reference's
base/IterativeRecommender.py with the per-user scoring + selection moved to the GPU.

Numeric contract kept from the reference:
  initModel   P = rand(m,k).astype(f32)/10 then Q (global NumPy stream, P drawn first)   (:36-39)
  isConverged NaN abort, the per-iteration print line, |delta| < 1e-3, bold driver          (:47-75)
  evalRanking mask train items, seed with first N candidates, overwrite-scan, lists file    (:77-173)
  ranking_performance  same scan on the first 300 test users, N=10, test items masked       (:175-235)
"""
from math import isnan

import numpy as np

from ..data.arrays import ArrayRecord, ranking_measure_ids
from ..evaluation.measure import Measure
from ..tool import config
from .recommender import Recommender


class IterativeRecommender(Recommender):
    def __init__(self, conf, trainingSet=None, testSet=None, fold='[1]'):
        super(IterativeRecommender, self).__init__(conf, trainingSet, testSet, fold)
        self.dev = None

    def readConfiguration(self):
        super(IterativeRecommender, self).readConfiguration()
        self.k = int(self.config['num.factors'])
        self.maxIter = int(self.config['num.max.iter'])
        rate = config.LineConfig(self.config['learnRate'])
        self.lRate = float(rate['-init'])
        self.maxLRate = float(rate['-max'])
        reg = config.LineConfig(self.config['reg.lambda'])
        self.regU, self.regI, self.regB = float(reg['-u']), float(reg['-i']), float(reg['-b'])

    def printAlgorConfig(self):
        super(IterativeRecommender, self).printAlgorConfig()
        print('Reduced Dimension:', self.k)
        print('Maximum Iteration:', self.maxIter)
        print('Regularization parameter: regU %.3f, regI %.3f, regB %.3f' % (self.regU, self.regI, self.regB))
        print('=' * 80)

    def initModel(self):
        self.P = np.random.rand(self.data.getSize('user'), self.k).astype(np.float32) / 10
        self.Q = np.random.rand(self.data.getSize(self.recType), self.k).astype(np.float32) / 10
        self.loss, self.lastLoss = 0, 0

    # the reference leaves these as stubs (:41-45); we fill them with a plain .npz of P and Q
    def saveModel(self):
        out = self.output['-dir'] if hasattr(self, 'output') else './'
        np.savez(out + self.config['recommender'] + self.foldInfo + '-factors.npz', P=self.P, Q=self.Q)

    def loadModel(self):
        out = self.output['-dir'] if hasattr(self, 'output') else './'
        with np.load(out + self.config['recommender'] + self.foldInfo + '-factors.npz', allow_pickle=False) as z:
            self.P, self.Q = z['P'], z['Q']

    def updateLearningRate(self, iter):
        if iter > 1:
            self.lRate *= 1.01 if abs(self.lastLoss) > abs(self.loss) else 0.5
        if self.maxLRate > 0 and self.lRate > self.maxLRate:
            self.lRate = self.maxLRate

    def _device(self):
        """HIP context of this process, created on first use (after any fork: yue.py -cv)."""
        if self.dev is None:
            from .._shim import Device
            gpu, true_topn = 0, False
            if self.config.contains('bpr.hip'):
                opts = config.LineConfig(self.config['bpr.hip'])
                if opts.contains('-gpu'):
                    gpu = int(opts['-gpu'])
                true_topn = opts.contains('-topn') and opts['-topn'] == 'true'
            self.dev = Device(gpu)
            if true_topn:        # a real top-N instead of the reference's overwrite-scan (off by default)
                self.dev.set_option('topn_true', 1)
        return self.dev

    def _sync_factors_to_device(self):
        dev = self._device()
        dev.set_factors(self.P, self.Q)
        arrays = self.data.to_arrays(self.recType)
        dev.set_interactions(arrays['indptr'], arrays['indices'], arrays['ev_ptr'], arrays['ev_i'])
        self._arrays = arrays
        self._device_factors_current = True

    def predict(self, u):
        'scores of all items for one user, item-id order (fp32)'
        u = self.data.getId(u, 'user')
        if not getattr(self, '_device_factors_current', False):
            self._sync_factors_to_device()
        return self.dev.scores(u)

    def isConverged(self, iter):
        if isnan(self.loss):
            print('Loss = NaN or Infinity: current settings does not fit the recommender! Change the settings and try again!')
            exit(-1)
        deltaLoss = (self.lastLoss - self.loss)
        print('%s %s iteration %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f'
              % (self.algorName, self.foldInfo, iter, self.loss, deltaLoss, self.lRate))
        converged = abs(deltaLoss) < 1e-3
        if not converged:
            self.updateLearningRate(iter)
        self.lastLoss = self.loss
        return converged

    def _scan(self, users, N, mask=None):
        """ids[nu, N] from the GPU selection kernel for user NAMES `users`."""
        if not getattr(self, '_device_factors_current', False):
            self._sync_factors_to_device()
        uids = np.array([self.data.getId(u, 'user') for u in users], np.int32)
        if mask is None:
            ids, _ = self.dev.topn_scan(uids, N)
        else:
            ids, _ = self.dev.topn_scan(uids, N, mask[0], mask[1])
        return ids

    def evalRanking(self):
        top = self._top_list()
        N = max(top)
        if N > 100 or N < 0:
            print('N can not be larger than 100! It has been reassigned with 10')
            N = 10
        if isinstance(self.data, ArrayRecord):
            return self._evalRanking_arrays(top, N)
        res = ['userId: recommendations in (itemId, ranking score) pairs, * means the item matches.\n']
        users = list(self.data.testSet.keys())
        ids = self._scan(users, N) if users else np.zeros((0, N), np.int32)
        names = self.data.id2name[self.recType]
        recList = {}
        userCount = len(users)
        for i, user in enumerate(users):
            recList[user] = [names[int(x)] for x in ids[i]]
            if i % 100 == 0:
                print(self.algorName, self.foldInfo, 'progress:' + str(i) + '/' + str(userCount))
            wanted = self.data.testSet[user]
            res.append(user + ':' + ''.join(item + '*' if item in wanted else item for item in recList[user]) + '\n')
        self._write_results(res, recList, top)

    def _evalRanking_arrays(self, top, N):
        """Array-native data: lists stay integer (``self.recUsers``, ``self.recIds``), the measures are
        computed on ids (data/arrays.py: ranking_measure_ids); beside the measure file the lists go out as integer
        arrays (``...-top-Nitems.npz``: users, ids) -- the reference's text line of concatenated names would be
        unreadable for numeric names."""
        from os.path import abspath
        from time import localtime, strftime, time
        from ..tool.file import FileIO
        d = self.data
        uids = d.testSet.user_ids().astype(np.int32)
        if not getattr(self, '_device_factors_current', False):
            self._sync_factors_to_device()
        self.recUsers = uids
        self.recIds = self.dev.topn_scan(uids, N)[0] if len(uids) else np.zeros((0, N), np.int32)
        self.measure = ranking_measure_ids(d.test_indptr, d.test_indices, uids, self.recIds, top, d.getSize(self.recType))
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        FileIO.writeFile(self.output['-dir'], self.config['recommender'] + '@' + stamp + '-measure' + self.foldInfo + '.txt', self.measure)
        if self.isOutput:
            np.savez(self.output['-dir'] + self.config['recommender'] + '@' + stamp + '-top-' + str(N) + 'items' + self.foldInfo + '.npz', users=uids, ids=self.recIds)
        print('The result has been output to ', abspath(self.output['-dir']), '.')
        print('The result of %s %s:\n%s' % (self.algorName, self.foldInfo, ''.join(self.measure)))

    def ranking_performance(self):
        N = 10
        itemcount = 0
        testSample = {}
        for user in self.data.testSet:
            itemcount += len(self.data.testSet[user])
            if len(testSample) == 300:
                break
            testSample[user] = self.data.testSet[user]
        users = list(testSample.keys())
        track_ids = self.data.name2id['track']
        rows = [np.sort(np.array([track_ids[item] for item in testSample[u]], np.int32)) for u in users]
        mp = np.zeros(len(users) + 1, np.int64)
        mp[1:] = np.cumsum([len(r) for r in rows])
        mi = np.concatenate(rows) if rows else np.zeros(0, np.int32)
        ids = self._scan(users, N, (mp, mi)) if users else np.zeros((0, N), np.int32)
        names = self.data.id2name['track']
        recList = {user: [names[int(x)] for x in ids[i]] for i, user in enumerate(users)}
        measure = Measure.rankingMeasure(testSample, recList, [10], itemcount)
        print('-' * 80)
        print('Ranking Performance ' + self.foldInfo + ' (Top-10 On 300 sampled users)')
        for m in measure[1:]:
            print(m.strip())
        print('-' * 80)
        return measure
