"""Ranking metrics over {user: [item names]} lists; same numbers and output strings as the
reference's evaluation/measure.py:7-66,91-101 (pinned by tests/golden/g7_measure.json)."""


class Measure(object):
    @staticmethod
    def hits(origin, res):
        return {user: len(set(origin[user]) & set(res[user])) for user in origin}

    @staticmethod
    def precision(hits, N):
        return float(sum(hits.values())) / (len(hits) * N)

    @staticmethod
    def recall(hits, origin):
        per_user = [float(hits[user]) / len(origin[user]) for user in hits]
        return sum(per_user) / float(len(per_user))

    @staticmethod
    def F1(prec, recall):
        return 2 * prec * recall / (prec + recall) if (prec + recall) != 0 else 0

    @staticmethod
    def MAP(origin, res, N):
        total = 0
        for user in res:
            found, acc = 0, 0
            for rank, item in enumerate(res[user]):
                if item in origin[user]:
                    found += 1
                    acc += found / (rank + 1.0)
            total += acc / (min(len(origin[user]), N) + 0.0)
        return total / len(res)

    @staticmethod
    def coverage(res, itemCount):
        seen = set()
        for user in res:
            seen.update(res[user])
        return len(seen) / float(itemCount)

    @staticmethod
    def rankingMeasure(origin, res, N, itemCount):
        print('rank measure...')
        out = []
        for n in N:
            cut = {user: res[user][:n] for user in res}
            if len(origin) != len(cut):
                print('The Lengths of test set and predicted set are not match!')
                exit(-1)
            hits = Measure.hits(origin, cut)
            prec = Measure.precision(hits, n)
            rec = Measure.recall(hits, origin)
            out.append('Top ' + str(n) + '\n')
            out.append('Precision:' + str(prec) + '\n')
            out.append('Recall:' + str(rec) + '\n')
            out.append('F1:' + str(Measure.F1(prec, rec)) + '\n')
            out.append('MAP:' + str(Measure.MAP(origin, cut, n)) + '\n')
            out.append('Coverage:' + str(Measure.coverage(cut, itemCount)) + '\n')
        return out
