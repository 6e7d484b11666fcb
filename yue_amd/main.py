"""Menu front end (reference main.py:7-80).  `python -m yue_amd.main` from the repository root."""
import time

from .tool.config import Config
from .yue import Yue

MENU = {'1': 'BPR', '2': 'FISM', 'a1': 'CUNE'}


def main():
    print('=' * 80)
    print('   Yue: Library for Music Recommendation (MI355X BPR path).   ')
    print('=' * 80)
    print('CF-based Recommenders:')
    print('1. BPR   2. FISM')
    print('Advanced Recommenders:')
    print('a1. CUNE (training loop; needs -friends, see recommender/advanced/CUNE.py)')
    print('=' * 80)
    order = input('Please enter the num of the algorithm to run it:')
    start = time.time()
    if order not in MENU:
        print('Error num!')
        exit(-1)
    conf = Config('./config/' + MENU[order] + '.conf')      # the user's own file, as in the reference (main.py:36-37)
    Yue(conf).execute()
    print("Run time: %f s" % (time.time() - start))


if __name__ == '__main__':
    main()
