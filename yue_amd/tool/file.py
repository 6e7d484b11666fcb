"""Text I/O with the behaviour of the reference's tool/file.py:10-52."""
import os
import re


class FileIO(object):
    @staticmethod
    def writeFile(dir, file, content, op='w'):
        if not os.path.exists(dir):
            os.makedirs(dir)
        with open(dir + file, op) as f:
            f.writelines(content)

    @staticmethod
    def deleteFile(filePath):
        if os.path.exists(filePath):
            os.remove(filePath)

    @staticmethod
    def loadDataSet(file, columns, binarized=False, threshold=3, delim=''):
        """list of {column: str} events; fields split on ``delim`` (default: comma, blank or tab)."""
        print('load dataset...')
        names = list(columns.keys())
        if len(names) < 2:
            print('The dataset needs more information or the record.setup setting has some problems...')
            exit(-1)
        where = [int(v) for v in columns.values()]
        splitter = re.compile(delim if delim != '' else ',| |\t')
        record = []
        with open(file) as f:
            for lineNo, line in enumerate(f, 1):
                fields = splitter.split(line.strip())
                try:
                    event = {}
                    for name, idx in zip(names, where):
                        event[name] = fields[idx]
                        if binarized and 'play' in event:
                            event['play'] = 1 if int(event['play']) >= threshold else 0
                except IndexError:
                    print('The record file is not in a correct format. Error Location: Line num %d' % lineNo)
                    exit(-1)
                record.append(event)
        return record
