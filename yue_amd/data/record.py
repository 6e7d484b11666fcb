"""Listening-log data model with the observable behaviour of the reference's data/record.py.

What the BPR path reads from it: ``name2id / id2name`` (ids in first-seen
order, keys of an event visited in ``-columns`` order, ``time`` skipped), ``userRecord`` (training
events per user, in training-set order), ``testSet`` (per user {item: count}, minus anything the
user has in training), ``getId / getSize``.  ``to_arrays()`` is ours: the CSR + event-list view the
HIP library uploads (include/yue_hip.h: yue_set_interactions).
"""
from collections import defaultdict

import numpy as np

from ..tool.config import LineConfig


class Record(object):
    'data access control'

    def __init__(self, config, trainingSet, testSet):
        self.config = config
        self.recordConfig = LineConfig(config['record.setup'])
        self.evalConfig = LineConfig(config['evaluation.setup'])
        self.name2id = defaultdict(dict)
        self.id2name = defaultdict(dict)
        self.listened = {kind: defaultdict(dict) for kind in ('artist', 'track', 'album')}
        self.artist2Album = defaultdict(dict)
        self.album2Track = defaultdict(dict)
        self.artist2Track = defaultdict(dict)
        self.Track2artist = defaultdict(dict)
        self.Track2album = defaultdict(dict)
        self.userRecord = defaultdict(list)
        self.trackRecord = defaultdict(list)
        self.testSet = defaultdict(dict)
        self.recordCount = 0
        self.columns = {}
        self.globalMean = 0
        self.userMeans = {}
        self.trackListen = {}
        self.PopTrack = {}
        # kept as the reference keeps it: assigned BEFORE any -byTime split
        self.trainingData = trainingSet

        for col in self.recordConfig['-columns'].split(','):
            name, pos = col.split(':')
            self.columns[name] = int(pos)
        if self.evalConfig.contains('-byTime'):
            trainingSet, testSet = self.splitDataByTime(trainingSet)
        self.preprocess(trainingSet, testSet)
        self.computePop(trainingSet)

    # reference data/record.py:108-123 -- per user, string-sorted by 'time', first int(len*(1-r)) train
    def splitDataByTime(self, dataset):
        ratio = float(self.evalConfig['-byTime'])
        per_user = defaultdict(list)
        for event in dataset:
            per_user[event['user']].append(event)
        train, test = [], []
        for user in per_user:
            ordered = sorted(per_user[user], key=lambda ev: ev['time'])
            cut = int(len(ordered) * (1 - ratio))
            train += ordered[:cut]
            test += ordered[cut:]
        return train, test

    def _register(self, entry):
        for key in entry:
            if key != 'time' and entry[key] not in self.name2id[key]:
                new_id = len(self.name2id[key])
                self.name2id[key][entry[key]] = new_id
                self.id2name[key][new_id] = entry[key]

    # reference data/record.py:138-202
    def preprocess(self, trainingSet, testSet):
        for entry in trainingSet:
            self.recordCount += 1
            self._register(entry)
            user = entry.get('user')
            if user is not None:
                self.userRecord[user].append(entry)
                for kind in ('artist', 'album', 'track'):
                    if kind in entry:
                        plays = self.listened[kind][entry[kind]]
                        plays[user] = plays.get(user, 0) + 1
            if 'artist' in entry and 'album' in entry:
                self.artist2Album[entry['artist']][entry['album']] = 1
            if 'album' in entry and 'track' in entry:
                self.album2Track[entry['album']] = self.name2id['track'][entry['track']]
                self.Track2album[entry['track']] = self.name2id['album'][entry['album']]
            if 'artist' in entry and 'track' in entry:
                self.artist2Track[entry['artist']] = self.name2id['track'][entry['track']]
                self.Track2artist[entry['track']] = self.name2id['artist'][entry['artist']]
            if 'track' in entry:
                self.trackRecord[entry['track']].append(entry)

        recType = self.evalConfig['-target']
        for entry in testSet:
            self._register(entry)
            if 'user' in entry:
                wanted = self.testSet[entry['user']]
                if recType in entry and entry[recType] not in wanted:
                    wanted[entry[recType]] = 1
                else:
                    wanted[entry[recType]] += 1

        # items a user has in training never count as test items
        for item, users in self.listened[recType].items():
            for user in users:
                if user in self.testSet:
                    self.testSet[user].pop(item, None)
                    if not self.testSet[user]:
                        del self.testSet[user]

    # reference data/record.py:125-135: total training plays of every track seen in `dataset`
    def computePop(self, dataset):
        print('computePop...')
        for event in dataset:
            if 'track' in event:
                total = sum(self.listened['track'][event['track']].values())
                if total > 0:
                    self.PopTrack[event['track']] = total
        print('computePop is finished...')
        print('PopTrack', len(self.PopTrack))

    def printTrainingSize(self):
        for kind in ('user', 'artist', 'album', 'track'):
            if kind in self.name2id:
                print(kind + ' count:', len(self.name2id[kind]))
        print('Training set size:', self.recordCount)

    def getId(self, obj, t):
        if obj in self.name2id[t]:
            return self.name2id[t][obj]
        print('No ' + t + ' ' + obj + ' exists!')
        exit(-1)

    def getSize(self, t):
        return len(self.name2id[t])

    def contains(self, obj, t):
        'whether the recType t is in trainging set'
        return obj in self.name2id[t]

    # ---- ours: the array view uploaded to the GPU ------------------------------------------
    def to_arrays(self, recType):
        """Events in userRecord order (recommender/cf/BPR.py:42-45) + sorted-unique listened rows
        (BPR.py:32-35).  userRecord iterates users in first-seen order, which is ascending user id
        because ids are handed out in the same pass; asserted here."""
        m = self.getSize('user')
        ev_ptr = np.zeros(m + 1, np.int64)
        ev_i = []
        last = -1
        for user, events in self.userRecord.items():
            uid = self.name2id['user'][user]
            assert uid > last, 'userRecord order must follow user ids'
            last = uid
            ev_ptr[uid + 1] = len(events)
            item_ids = self.name2id[recType]
            ev_i.extend(item_ids[ev[recType]] for ev in events)
        ev_ptr = np.cumsum(ev_ptr)
        ev_i = np.asarray(ev_i, dtype=np.int32)
        ev_u = np.repeat(np.arange(m, dtype=np.int64), np.diff(ev_ptr))
        n = self.getSize(recType)
        keys = np.unique(ev_u * n + ev_i)
        indptr = np.zeros(m + 1, np.int64)
        np.add.at(indptr, keys // n + 1, 1)
        return {'ev_ptr': ev_ptr, 'ev_i': ev_i, 'indptr': np.cumsum(indptr), 'indices': (keys % n).astype(np.int32)}
