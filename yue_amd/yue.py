"""Driver: config -> dataset -> recommender class by name -> run (reference yue.py:10-135).

Differences from the reference, all outside the numeric path: the class is imported with
importlib instead of exec/eval, `mkl` is not needed, and only recommenders that exist in
yue_amd.recommender are importable (BPR and FISM in this build).

One addition: ``record.setup=-format csr`` makes ``record`` a binary integer data set
(yue_amd/data/arrays.py: save_csr / load_csr) that carries its own held-out items; it goes to the
recommender as an ArrayRecord, whatever ``evaluation.setup`` says about splitting (only ``-cv`` is refused).
"""
import importlib
from multiprocessing import Manager, Process
from random import random
from time import localtime, strftime, time

from .tool.config import LineConfig
from .tool.file import FileIO


def _holdout(events, test_ratio):
    """-ap: every event goes to the test side with probability test_ratio (reference tool/dataSplit.py:9-23)."""
    if not 0 < test_ratio < 1:
        test_ratio = 0.3
    sides = ([], [])
    for event in events:
        sides[random() < test_ratio].append(event)
    return sides


def _folds(events, k):
    """-cv: fold f tests the events whose position is f modulo k (reference tool/dataSplit.py:26-37)."""
    for f in range(k):
        yield events[:0] + [e for pos, e in enumerate(events) if pos % k != f], events[f::k]


def _find_recommender(name):
    last = None
    for family in ('baseline', 'cf', 'advanced'):
        try:
            module = importlib.import_module('yue_amd.recommender.%s.%s' % (family, name))
            return getattr(module, name)
        except ImportError as err:
            last = err
    raise ImportError('recommender %s is not part of this build (%s)' % (name, last))


class Yue(object):
    def __init__(self, config):
        self.trainingData = []
        self.testData = []
        self.measure = []
        self.config = config
        setup = LineConfig(config['record.setup'])
        if not self.config.contains('evaluation.setup'):
            print('Evaluation is not well configured!')
            exit(-1)
        self.evaluation = LineConfig(config['evaluation.setup'])
        if setup.contains('-format') and setup['-format'] == 'csr':
            if self.evaluation.contains('-cv'):
                print('-cv needs the text log: a csr data set carries its own held-out items.')
                exit(-1)
            from .data.arrays import load_csr
            target = self.evaluation['-target'] if self.evaluation.contains('-target') else 'track'
            self.trainingData = load_csr(config['record'], target)
            print('preprocessing...')
            return
        columns = {}
        for col in setup['-columns'].split(','):
            name, pos = col.split(':')
            columns[name] = int(pos)
        delim = setup['-delim'] if setup.contains('-delim') else ''

        binarized = self.evaluation.contains('-b')
        bottom = float(self.evaluation['-b']) if binarized else 0

        def load(path):
            return FileIO.loadDataSet(path, columns=columns, binarized=binarized, threshold=bottom, delim=delim)

        if self.evaluation.contains('-testSet'):
            self.trainingData = load(config['record'])
            self.testData = load(self.evaluation['-testSet'])
        elif self.evaluation.contains('-ap'):
            self.trainingData, self.testData = _holdout(load(config['record']), float(self.evaluation['-ap']))
        elif self.evaluation.contains('-byTime'):
            self.trainingData = load(config['record'])      # Record splits per user by time
            self.testData = []
        elif self.evaluation.contains('-cv'):
            self.trainingData = load(config['record'])
        print('preprocessing...')

    def execute(self):
        cls = _find_recommender(self.config['recommender'])
        if not self.evaluation.contains('-cv'):
            cls(self.config, self.trainingData, self.testData).execute()
            return
        k = int(self.evaluation['-cv'])
        if k <= 1 or k > 10:
            k = 3
        shared = Manager().dict()
        tasks = []
        for order, (train, test) in enumerate(_folds(self.trainingData, k), 1):
            # the HIP context is created inside the child (first device call), never before the fork
            tasks.append(Process(target=run, args=(shared, cls(self.config, train, test, '[' + str(order) + ']'), order)))
        parallel = self.evaluation.contains('-p')
        for p in tasks:
            p.start()
            if not parallel:
                p.join()
        if parallel:
            for p in tasks:
                p.join()
        self.measure = [dict(shared)[i] for i in range(1, k + 1)]
        res = []
        for pos, text in enumerate(self.measure[0]):
            if text[:3] == 'Top':
                res.append(text)
                continue
            label = text.split(':')[0]
            mean = sum(float(self.measure[fold][pos].split(':')[1]) for fold in range(k)) / k
            res.append(label + ':' + str(mean) + '\n')
        stamp = strftime("%Y-%m-%d %H-%M-%S", localtime(time()))
        outDir = LineConfig(self.config['output.setup'])['-dir']
        FileIO.writeFile(outDir, self.config['recommender'] + '@' + stamp + '-' + str(k) + '-fold-cv' + '.txt', res)
        print('The result of %d-fold cross validation:\n%s' % (k, ''.join(res)))


def run(measure, algor, order):
    measure[order] = algor.execute()
