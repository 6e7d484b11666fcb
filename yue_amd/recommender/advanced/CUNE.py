#coding:utf8
"""CUNE's training loop (collaborative user network embedding, two-level BPR) behind the reference's plugin hooks.

Replaces the NumPy SGD loop of the reference's recommender/advanced/CUNE.py:120-178 with the device function
yue_cune_steps (include/yue_hip.h): per (user, positive item) three steps, each either the two-level step with a
friends' item k and a sampled negative j or, for a user without friends' items, the plain (i over j) step.  Draws are
made on the host exactly as the reference makes them (``choice`` over the friends' items / the item names, rejection of
listened items through ``data.listened``), so a seeded ``random`` gives the reference's stream; the loss is added up as
the reference adds it (the regulariser once per user, :175 -- the running loss is a float32 from then on).

What is NOT here: the stage that produces the friends' item sets -- the collaborative user network, the random walks and
the Word2Vec user embedding (CUNE.py:34-118; gensim is absent in this build, parity unpinned).  The sets come from
``CUNE=... -friends FILE`` (one line per user: ``user:friend,friend,...``; a user's friends' items are the items its
friends listened to and the user did not, CUNE.py:115-117) or, when an ``IPositiveSet`` attribute is set on the instance
beforehand, from there; without either every user takes the plain branch.
"""
from collections import defaultdict
from math import isnan
from random import choice

import numpy as np

from ...base.IterativeRecommender import IterativeRecommender
from ...tool.config import LineConfig


class CUNE(IterativeRecommender):

    def __init__(self, conf, trainingSet=None, testSet=None, fold='[1]'):
        super(CUNE, self).__init__(conf, trainingSet, testSet, fold)

    def readConfiguration(self):
        super(CUNE, self).readConfiguration()
        options = LineConfig(self.config['CUNE'])
        self.walkCount = int(options['-T'])
        self.walkLength = int(options['-L'])
        self.walkDim = int(options['-l'])
        self.winSize = int(options['-w'])
        self.topK = int(options['-k'])
        self.s = float(options['-s'])
        self.epoch = int(options['-ep'])
        self.friendsFile = options['-friends'] if options.contains('-friends') else ''

    def _item_sets(self):
        self.PositiveSet = defaultdict(list)                    # CUNE.py:107-113
        for user in self.data.userRecord:
            for event in self.data.userRecord[user]:
                self.PositiveSet[user].append(event[self.recType])
        if hasattr(self, 'IPositiveSet'):
            return
        self.IPositiveSet = defaultdict(list)
        if not self.friendsFile:
            print('CUNE: no -friends file: the user-network stage (gensim) is not part of this build; every user takes the plain step.')
            return
        for line in open(self.friendsFile):
            if ':' not in line:
                continue
            user, friends = line.strip().split(':', 1)
            for friend in [f for f in friends.split(',') if f]:
                if user in self.PositiveSet and friend in self.PositiveSet:      # :115-117
                    self.IPositiveSet[user] += list(set(self.PositiveSet[friend]).difference(self.PositiveSet[user]))

    def buildModel(self):
        self._item_sets()
        d, rt = self.data, self.recType
        self._sync_factors_to_device()
        dev = self.dev
        print('Training...')
        iteration = 0
        while iteration < self.maxIter:
            self.loss = 0
            itemList = list(d.name2id[rt].keys())
            for user in self.PositiveSet:
                u = d.getId(user, 'user')
                uu, ii, kk, jj = [], [], [], []
                for item in self.PositiveSet[user]:
                    i = d.getId(item, rt)
                    for _n in range(3):
                        k = -1
                        if len(self.IPositiveSet[user]) > 0:
                            k = d.getId(choice(self.IPositiveSet[user]), rt)
                        item_j = choice(itemList)
                        while user in d.listened[rt][item_j]:
                            item_j = choice(itemList)
                        uu.append(u); ii.append(i); kk.append(k); jj.append(d.getId(item_j, rt))
                for x in dev.cune_steps(uu, ii, kk, jj, self.s, self.lRate, self.regU, self.regI):
                    self.loss += float(x)
                sumP, sumQ = dev.sumsq()
                # :175 inside the user loop: NumPy float32 sums times python floats -> the loss turns float32
                self.loss += self.regU * np.float32(sumP) + self.regI * np.float32(sumQ)
                if isnan(float(self.loss)):
                    break
            iteration += 1
            if self.isConverged(iteration):
                break
        self.P, self.Q = dev.get_factors(self.P, self.Q)
        self._device_factors_current = True
