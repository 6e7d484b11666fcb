"""Plugin template: same lifecycle and hooks as the reference's base/recommender.py.

execute(): readConfiguration -> printAlgorConfig (fold [1]) -> loadModel | initModel + buildModel
-> evalRanking -> saveModel, returns self.measure (reference base/recommender.py:152-174).
Constructor signature of every plugin: (conf, trainingSet=None, testSet=None, fold='[1]').
"""
from collections import defaultdict
from os.path import abspath
from time import localtime, strftime, time

from ..data.record import Record
from ..evaluation.measure import Measure
from ..tool.config import LineConfig
from ..tool.file import FileIO


class Recommender(object):
    def __init__(self, conf, trainingSet=None, testSet=None, fold='[1]'):
        self.config = conf
        self.isSaveModel = False
        self.isLoadModel = False
        self.isOutput = True
        self.data = Record(self.config, trainingSet, testSet)
        self.foldInfo = fold
        self.evalConfig = LineConfig(self.config['evaluation.setup'])
        self.recType = self.evalConfig['-target'] if self.evalConfig.contains('-target') else 'track'
        if self.evalConfig.contains('-cold'):
            self._keep_cold_items(int(self.evalConfig['-cold']))
        if self.evalConfig.contains('-sample'):
            users = list(self.data.testSet.keys())
            for user in users[:int(len(users) * 0.9)]:
                del self.data.testSet[user]

    def _keep_cold_items(self, threshold):
        # reference base/recommender.py:22-39: drop test items with more than `threshold` training plays
        drop = defaultdict(list)
        for user in self.data.testSet:
            if user in self.data.userRecord:
                for item in self.data.testSet[user]:
                    if len(self.data.trackRecord[item]) > threshold:
                        drop[user].append(item)
        for user, items in drop.items():
            for item in items:
                del self.data.testSet[user][item]
            if not self.data.testSet[user]:
                del self.data.testSet[user]

    def readConfiguration(self):
        self.algorName = self.config['recommender']
        self.output = LineConfig(self.config['output.setup'])
        self.isOutput = self.output.isMainOn()
        self.ranking = LineConfig(self.config['item.ranking'])

    def printAlgorConfig(self):
        "show algorithm's configuration"
        print('Algorithm:', self.config['recommender'])
        print('Training set:', abspath(self.config['record']))
        if self.evalConfig.contains('-testSet'):
            print('Test set:', abspath(self.evalConfig.getOption('-testSet')))
        self.data.printTrainingSize()
        print('=' * 80)

    def initModel(self):
        pass

    def buildModel(self):
        'build the model (for model-based algorithms )'
        pass

    def saveModel(self):
        pass

    def loadModel(self):
        pass

    def predict(self, user):
        return []

    def _top_list(self):
        top = [int(num) for num in self.ranking['-topN'].split(',')]
        return top

    def _write_results(self, res, recList, top):
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        outDir = self.output['-dir']
        if self.isOutput:
            fileName = ''
            if self.ranking.contains('-topN'):
                fileName = self.config['recommender'] + '@' + stamp + '-top-' + self.ranking['-topN'] + 'items' + self.foldInfo + '.txt'
            FileIO.writeFile(outDir, fileName, res)
            print('The result has been output to ', abspath(outDir), '.')
        fileName = self.config['recommender'] + '@' + stamp + '-measure' + self.foldInfo + '.txt'
        self.measure = Measure.rankingMeasure(self.data.testSet, recList, top, self.data.getSize(self.recType))
        FileIO.writeFile(outDir, fileName, self.measure)
        print('The result of %s %s:\n%s' % (self.algorName, self.foldInfo, ''.join(self.measure)))

    def evalRanking(self):
        """Generic form for recommenders whose predict() returns an ordered item list
        (reference base/recommender.py:85-150)."""
        top = self._top_list()
        N = int(top[-1])
        if N > 100 or N < 0:
            print('N can not be larger than 100! It has been reassigned with 10')
            N = 10
        res = ['userId: recommendations in (itemId, ranking score) pairs, * means the item matches, $ means the unpop item\n']
        recList = {}
        userCount = len(self.data.testSet)
        for i, user in enumerate(self.data.testSet):
            ranked = self.predict(user) if user in self.data.userRecord else ['0'] * N
            position = {}
            for k, item in enumerate(ranked):
                position[item] = k
            for event in self.data.userRecord[user]:
                position.pop(event[self.recType], None)
            recList[user] = [item for item, _ in sorted(position.items(), key=lambda d: d[1])][:N]
            if i % 100 == 0:
                print(self.algorName, self.foldInfo, 'progress:' + str(i) + '/' + str(userCount))
            line = user + ':'
            for item in recList[user]:
                if item in self.data.testSet[user]:
                    line += '*'
                if item in self.data.PopTrack:
                    line += '$'
                line += item + ','
            res.append(line + '\n')
        self._write_results(res, recList, top)

    def execute(self):
        self.readConfiguration()
        if self.foldInfo == '[1]':
            self.printAlgorConfig()
        if self.isLoadModel:
            print('Loading model %s...' % (self.foldInfo))
            self.loadModel()
        else:
            print('Initializing model %s...' % (self.foldInfo))
            self.initModel()
            print('Building Model %s...' % (self.foldInfo))
            self.buildModel()
        print('Predicting %s...' % (self.foldInfo))
        self.evalRanking()
        if self.isSaveModel:
            print('Saving model %s...' % (self.foldInfo))
            self.saveModel()
        return self.measure
