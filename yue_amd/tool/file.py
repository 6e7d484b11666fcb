"""Text I/O for listening logs and result files.

Same observable behaviour as the reference's ``FileIO`` (tool/file.py:10-52): ``loadDataSet``
turns every line of a log into a ``{column: string}`` event using the ``-columns`` map of
``record.setup``; ``writeFile`` writes a list of strings under ``dir + file``.
"""
import os
import re

_DEFAULT_DELIMITERS = ',| |\t'


def _parse_line(fields, names, positions, binarized, threshold):
    event = {}
    for name, pos in zip(names, positions):
        event[name] = fields[pos]
        if binarized and 'play' in event:
            event['play'] = int(int(event['play']) >= threshold)
    return event


def read_events(path, columns, binarized=False, threshold=3, delim=''):
    names = list(columns)
    if len(names) < 2:
        print('The dataset needs more information or the record.setup setting has some problems...')
        exit(-1)
    positions = [int(columns[name]) for name in names]
    split = re.compile(delim or _DEFAULT_DELIMITERS).split
    events = []
    with open(path) as handle:
        for number, text in enumerate(handle, start=1):
            try:
                events.append(_parse_line(split(text.strip()), names, positions, binarized, threshold))
            except IndexError:
                print('The record file is not in a correct format. Error Location: Line num %d' % number)
                exit(-1)
    return events


class FileIO(object):
    """Namespace kept for drop-in use: FileIO.loadDataSet / writeFile / deleteFile."""

    @staticmethod
    def loadDataSet(file, columns, binarized=False, threshold=3, delim=''):
        print('load dataset...')
        return read_events(file, columns, binarized, threshold, delim)

    @staticmethod
    def writeFile(dir, file, content, op='w'):
        os.makedirs(dir, exist_ok=True)
        with open(dir + file, op) as handle:
            handle.writelines(content)

    @staticmethod
    def deleteFile(filePath):
        if os.path.exists(filePath):
            os.remove(filePath)
