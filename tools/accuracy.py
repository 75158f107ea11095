#!/usr/bin/env python
"""Hit-rate evaluator with the reference's metric definitions (tools/accuracy.py:34-45):
    python tools/accuracy.py <expected.csv> <result_detail.csv>
expected.csv has columns query,answer,time (genquery.py writes it); a query counts as
"song correct" when the predicted file's basename equals the expected one, "near" when in
addition |time error| <= 0.5 s, "exact" when <= 0.25 s.  Prints the three reference lines
and returns the numbers."""
import csv
import os
import sys


def evaluate(groundtruth, predict):
    with open(groundtruth, "r", newline="") as fin:
        gt = {os.path.basename(r["query"]): r for r in csv.DictReader(fin)}
    total = correct = near = exact = 0
    with open(predict, "r", newline="") as fin:
        for row in csv.DictReader(fin):
            want = gt[os.path.basename(row["query"])]
            total += 1
            if os.path.basename(want["answer"]) == os.path.basename(row["answer"]):
                correct += 1
                err = abs(float(want["time"]) - float(row["time"]))
                exact += err <= 0.25
                near += err <= 0.5
    return dict(total=total, song=correct, near=near, exact=exact)


def main(argv):
    if len(argv) < 3:
        print("Usage: python %s <groundtruth csv> <predict detail csv>" % argv[0])
        return 1
    r = evaluate(argv[1], argv[2])
    t = max(r["total"], 1)
    print("exact match correct %d acc %.2f" % (r["exact"], r["exact"] / t * 100))
    print("near match correct %d acc %.2f" % (r["near"], r["near"] / t * 100))
    print("song correct %d acc %.2f" % (r["song"], r["song"] / t * 100))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
