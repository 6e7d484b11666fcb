"""Configuration readers with the semantics of the reference's tool/config.py:3-88.

``Config``      one ``key=value`` per line, exactly one '=' (malformed lines only warn);
                a missing key prints and exits (-1), as the reference does.
``LineConfig``  ``[on|off] -opt value ... -opt2 ...`` mini language.  A token is an option when
                it starts with '-' and its tail is not all digits (so ``-0.5`` IS an option and
                ``-1`` is a value -- quirk kept, pinned by tests/golden/g1_config.json).
"""
import os


def _die(msg):
    print(msg)
    exit(-1)


class Config(object):
    def __init__(self, fileName):
        self.config = {}
        self.readConfiguration(fileName)

    def contains(self, key):
        return key in self.config

    def __getitem__(self, item):
        if item not in self.config:
            _die('parameter ' + item + ' is invalid!')
        return self.config[item]

    getOptions = __getitem__

    def readConfiguration(self, fileName):
        path = os.path.abspath(fileName)
        if not os.path.exists(path):
            print('config file is not found!')
            raise IOError
        with open(path) as f:
            for lineno, raw in enumerate(f):
                text = raw.strip()
                if not text:
                    continue
                parts = text.split('=')
                if len(parts) != 2:
                    print('config file is not in the correct format! Error Line:%d' % lineno)
                    continue
                self.config[parts[0]] = parts[1]


def _is_option(token):
    return token.startswith('-') and not token[1:].isdigit()


class LineConfig(object):
    def __init__(self, content):
        self.line = content.strip().split(' ')
        self.mainOption = self.line[0] == 'on'
        self.options = {}
        tokens = self.line
        for pos, token in enumerate(tokens):
            if not _is_option(token):
                continue
            stop = pos + 1
            while stop < len(tokens) and not _is_option(tokens[stop]):
                stop += 1
            self.options[token] = ' '.join(tokens[pos + 1:stop])

    def contains(self, key):
        return key in self.options

    def __getitem__(self, item):
        if item not in self.options:
            _die('parameter ' + item + ' is invalid!')
        return self.options[item]

    getOption = __getitem__

    def isMainOn(self):
        return self.mainOption
