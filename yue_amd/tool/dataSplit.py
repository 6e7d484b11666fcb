"""Hold-out and k-fold splits (reference tool/dataSplit.py:9-37); no arithmetic on the hot path."""
from random import random


class DataSplit(object):
    @staticmethod
    def dataSplit(data, test_ratio=0.3, output=False, path='./', order=1):
        if not 0 < test_ratio < 1:
            test_ratio = 0.3
        train, test = [], []
        for entry in data:
            (test if random() < test_ratio else train).append(entry)
        if output:
            from .file import FileIO
            FileIO.writeFile(path, 'testSet[' + str(order) + ']', test)
            FileIO.writeFile(path, 'trainingSet[' + str(order) + ']', train)
        return train, test

    @staticmethod
    def crossValidation(data, k):
        if k <= 1 or k > 10:
            k = 3
        for fold in range(k):
            yield ([e for pos, e in enumerate(data) if pos % k != fold],
                   [e for pos, e in enumerate(data) if pos % k == fold])
