"""Plugin template of the Yue driver, hosted for the MI355X BPR path.

What a plugin is (contract kept from the reference, base/recommender.py:10-174, so that an existing
recommender class drops in):

  constructor   ``(conf, trainingSet=None, testSet=None, fold='[1]')``; builds ``self.data`` (Record),
                reads ``evaluation.setup`` (``-target`` type, ``-cold N`` and ``-sample`` test filters)
  hooks         ``readConfiguration  printAlgorConfig  initModel  buildModel  saveModel  loadModel
                predict  evalRanking`` -- subclasses override what they need
  execute()     readConfiguration -> printAlgorConfig (fold [1] only) -> loadModel | initModel +
                buildModel -> evalRanking -> [saveModel]; returns ``self.measure`` (list of strings)

Ours on top of that: ``_top_list`` / ``_write_results`` (shared by the factor models' GPU ranking)
and ``test_user_names``.
"""
from collections import defaultdict
from os.path import abspath
from time import localtime, strftime, time

from ..data.arrays import ArrayRecord
from ..data.record import Record
from ..evaluation.measure import Measure
from ..tool.config import LineConfig
from ..tool.file import FileIO

_RULE = '=' * 80


class Recommender(object):
    def __init__(self, conf, trainingSet=None, testSet=None, fold='[1]'):
        self.config = conf
        self.foldInfo = fold
        self.isSaveModel = self.isLoadModel = False
        self.isOutput = True
        # an ArrayRecord (integer arrays) is taken as it is; text-log events go through Record
        self.data = trainingSet if isinstance(trainingSet, ArrayRecord) else Record(conf, trainingSet, testSet)
        self.evalConfig = LineConfig(conf['evaluation.setup'])
        self.recType = self.evalConfig['-target'] if self.evalConfig.contains('-target') else 'track'
        if self.evalConfig.contains('-cold'):
            self._restrict_to_cold_items(int(self.evalConfig['-cold']))
        if self.evalConfig.contains('-sample'):
            self._keep_last_tenth_of_test_users()

    # ---- test-set filters -----------------------------------------------------------------
    def _restrict_to_cold_items(self, max_plays):
        """``-cold N``: test items played more than N times in training are dropped
        (only for users that also have training records); users left without items go too."""
        doomed = defaultdict(list)
        for user, wanted in self.data.testSet.items():
            if user in self.data.userRecord:
                doomed[user] = [item for item in wanted if len(self.data.trackRecord[item]) > max_plays]
        for user, items in doomed.items():
            for item in items:
                del self.data.testSet[user][item]
            if not self.data.testSet[user]:
                del self.data.testSet[user]

    def _keep_last_tenth_of_test_users(self):
        """``-sample``: the first 90 % of the test users (dict order) are not evaluated."""
        names = self.test_user_names()
        for user in names[:int(len(names) * 0.9)]:
            del self.data.testSet[user]

    def test_user_names(self):
        return list(self.data.testSet.keys())

    # ---- hooks ----------------------------------------------------------------------------
    def readConfiguration(self):
        conf = self.config
        self.algorName = conf['recommender']
        self.output = LineConfig(conf['output.setup'])
        self.isOutput = self.output.isMainOn()
        self.ranking = LineConfig(conf['item.ranking'])

    def printAlgorConfig(self):
        "show algorithm's configuration"
        lines = [('Algorithm:', self.config['recommender']), ('Training set:', abspath(self.config['record']))]
        if self.evalConfig.contains('-testSet'):
            lines.append(('Test set:', abspath(self.evalConfig.getOption('-testSet'))))
        for label, value in lines:
            print(label, value)
        self.data.printTrainingSize()
        print(_RULE)

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

    def evalRanking(self):
        """Ranking evaluation.  The reference's base class ranks recommenders whose predict() returns
        an ordered item list (base/recommender.py:85-150); none of those is part of this build, the
        factor models override this hook (IterativeRecommender.evalRanking)."""
        raise NotImplementedError('list-returning recommenders are outside the MI355X BPR build')

    # ---- shared by the ranking evaluations --------------------------------------------------
    def _top_list(self):
        return [int(num) for num in self.ranking['-topN'].split(',')]

    def _write_results(self, res, recList, top):
        """Lists file (when output is on), measure file, ``self.measure`` and the final print --
        file names as in the reference: ``<Algo>@<time>-top-<N>items<fold>.txt`` / ``-measure<fold>.txt``."""
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        outDir = self.output['-dir']
        prefix = self.config['recommender'] + '@' + stamp
        if self.isOutput:
            listing = prefix + '-top-' + self.ranking['-topN'] + 'items' + self.foldInfo + '.txt' if self.ranking.contains('-topN') else ''
            FileIO.writeFile(outDir, listing, res)
            print('The result has been output to ', abspath(outDir), '.')
        self.measure = Measure.rankingMeasure(self.data.testSet, recList, top, self.data.getSize(self.recType))
        FileIO.writeFile(outDir, prefix + '-measure' + self.foldInfo + '.txt', self.measure)
        print('The result of %s %s:\n%s' % (self.algorName, self.foldInfo, ''.join(self.measure)))

    # ---- lifecycle ------------------------------------------------------------------------
    def execute(self):
        fold = self.foldInfo
        self.readConfiguration()
        if fold == '[1]':
            self.printAlgorConfig()
        if self.isLoadModel:
            steps = [('Loading model %s...', self.loadModel)]
        else:
            steps = [('Initializing model %s...', self.initModel), ('Building Model %s...', self.buildModel)]
        steps.append(('Predicting %s...', self.evalRanking))
        if self.isSaveModel:
            steps.append(('Saving model %s...', self.saveModel))
        for message, hook in steps:
            print(message % fold)
            hook()
        return self.measure
