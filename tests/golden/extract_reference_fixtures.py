"""Turns the known-answer cases held by the reference's OWN unit tests into data fixtures (boards + expected values).

Run in the build container only (reads /root/reference/test/...).  Output: tests/golden/ref_rules_cases.json,
ref_features_cases.json, ref_movegen_cases.json.  Only inputs and expected outputs are stored — no reference source.

Sources: test/game/test_{freestyle,standard,caro,renju}.cpp (getOutcome / isForbidden),
test/networks/test_NNInputFeatures.cpp (bit layout), test/search/alpha_beta/test_move_generator.cpp (37 cases).
"""
import json
import os
import re

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))

SIGN = {"_": 0, "!": 0, "?": 0, "X": 1, "O": 2}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    lines = []
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("//"):
            continue
        lines.append(line)
    return "\n".join(lines)


def split_tests(text):
    """yields (name, body) for every live TEST/TEST_F"""
    for m in re.finditer(r"TEST(?:_F)?\((\w+),\s*(\w+)\)\s*\{", text):
        start = m.end()
        depth, i = 1, start
        while depth > 0:
            if text[i] == "{":
                depth += 1
            elif text[i] == "}":
                depth -= 1
            i += 1
        yield m.group(1) + "." + m.group(2), text[start:i - 1]


def parse_board(block):
    rows = re.findall(r'"([^"]*)\\n"', block)
    board = []
    for r in rows:
        board.append([SIGN[c] for c in r if c not in " "])
    assert all(len(r) == len(board[0]) for r in board), block
    return board


def parse_move(txt):
    return dict(sign=SIGN[txt[0]], col=ord(txt[1]) - ord("a"), row=int(txt[2:]))


def find_boards(body):
    """returns list of (position_in_body, board)"""
    out = []
    for m in re.finditer(r'(?:Board::fromString|set_board)\(\s*((?:"[^"]*"\s*)+)\)', body):
        out.append((m.start(), parse_board(m.group(1))))
    return out


def rules_cases():
    cases = []
    for fname in ["test_freestyle.cpp", "test_standard.cpp", "test_caro.cpp", "test_renju.cpp"]:
        text = strip_comments(open(os.path.join(REF, "game", fname)).read())
        for name, body in split_tests(text):
            boards = find_boards(body)
            if not boards:
                continue
            # walk the statements in source order so that add_move()/undo_move() between expectations are honoured
            events = []
            for m in re.finditer(r'EXPECT_EQ\(getOutcome\(GameRules::(\w+),\s*board,\s*Move\("(\w+)"\)\),\s*GameOutcome::(\w+)\)', body):
                events.append((m.start(), "outcome", m))
            for m in re.finditer(r'EXPECT_(TRUE|FALSE)\(is_forbidden\(Move\("(\w+)"\)\)\)', body):
                events.append((m.start(), "forbidden", m))
            for m in re.finditer(r'\b(add_move|undo_move)\(Move\("(\w+)"\)\)', body):
                events.append((m.start(), m.group(1), m))
            for p, b in boards:
                events.append((p, "board", b))
            events.sort(key=lambda e: e[0])
            current, checks = None, []

            def flush():
                if current is not None and checks:
                    cases.append(dict(name=fname + ":" + name + ("#%d" % len(cases)), board=[list(r) for r in current], checks=list(checks)))

            for pos, kind, m in events:
                if kind == "board":
                    flush()
                    current, checks = [list(r) for r in m], []
                elif kind in ("add_move", "undo_move"):
                    flush()
                    mv = parse_move(m.group(2))
                    current = [list(r) for r in current]
                    current[mv["row"]][mv["col"]] = mv["sign"] if kind == "add_move" else 0
                    checks = []
                elif kind == "outcome":
                    checks.append(dict(kind="outcome", rules=m.group(1), move=parse_move(m.group(2)), expected=m.group(3)))
                else:
                    checks.append(dict(kind="forbidden", move=parse_move(m.group(2)), expected=(m.group(1) == "TRUE")))
            flush()
    return cases


def features_cases():
    text = strip_comments(open(os.path.join(REF, "networks", "test_NNInputFeatures.cpp")).read())
    cases = []
    for name, body in split_tests(text):
        boards = find_boards(body)
        if not boards or "augment" in name:
            continue
        sign = re.search(r"sign_to_move = Sign::(\w+)", body).group(1)
        rules = re.search(r"GameConfig cfg\(GameRules::(\w+)", body).group(1)
        bits = []
        for m in re.finditer(r"EXPECT_(TRUE|FALSE)\(is_set_bit<(\d+)>\(features\.at\((\d+),\s*(\d+)\)\)\)", body):
            bits.append(dict(row=int(m.group(3)), col=int(m.group(4)), bit=int(m.group(2)), expected=(m.group(1) == "TRUE")))
        cases.append(dict(name=name, board=boards[0][1], sign_to_move=sign, rules=rules, bits=bits))
    return cases


def movegen_cases():
    text = strip_comments(open(os.path.join(REF, "search", "alpha_beta", "test_move_generator.cpp")).read())
    cases = []
    for name, body in split_tests(text):
        boards = find_boards(body)
        if not boards:
            continue
        wrappers = {}
        for m in re.finditer(r"MoveGenWrapper (\w+)\(GameRules::(\w+),\s*board,\s*Sign::(\w+)\)", body):
            board = [b for p, b in boards if p < m.start()][-1]
            wrappers[m.group(1)] = dict(rules=m.group(2), sign=m.group(3), board=board)
        lists = {}
        for m in re.finditer(r"ActionList (\w+) = (\w+)\(MoveGeneratorMode::(\w+)\)", body):
            w = wrappers[m.group(2)]
            lists[m.group(1)] = dict(rules=w["rules"], sign_to_move=w["sign"], board=w["board"], mode=m.group(3),
                                     size=None, must_defend=None, has_initiative=None, contains=[], not_contains=[], scores=[])
        for m in re.finditer(r"EXPECT_EQ\((\w+)\.size\(\),\s*(\d+)\)", body):
            lists[m.group(1)]["size"] = int(m.group(2))
        for m in re.finditer(r"EXPECT_(TRUE|FALSE)\((\w+)\.(must_defend|has_initiative)\)", body):
            lists[m.group(2)][m.group(3)] = (m.group(1) == "TRUE")
        for m in re.finditer(r'EXPECT_(TRUE|FALSE)\((\w+)\.contains\(Move\("(\w+)"\)\)\)', body):
            key = "contains" if m.group(1) == "TRUE" else "not_contains"
            lists[m.group(2)][key].append(parse_move(m.group(3)))
        for m in re.finditer(r'EXPECT_EQ\((\w+)\.getScoreOf\(Move\("(\w+)"\)\),\s*Score::(\w+)\((\d+)\)\)', body):
            lists[m.group(1)]["scores"].append(dict(move=parse_move(m.group(2)), kind=m.group(3), n=int(m.group(4))))
        for m in re.finditer(r"EXPECT_TRUE\((\w+)\.equals\((\w+)\)\)", body):
            lists[m.group(1)]["equals"] = name + ":" + m.group(2)
        for lname, l in lists.items():
            cases.append(dict(name=name + ":" + lname, **l))
        unparsed = [ln.strip() for ln in body.split("\n") if "EXPECT" in ln and not re.search(
            r"\.size\(\)|\.must_defend|\.has_initiative|\.contains\(|\.getScoreOf\(|\.equals\(", ln)]
        if unparsed:
            cases[-1]["unparsed_expectations"] = unparsed
    return cases


if __name__ == "__main__":
    r = rules_cases()
    f = features_cases()
    m = movegen_cases()
    json.dump(r, open(os.path.join(OUT, "ref_rules_cases.json"), "w"))
    json.dump(f, open(os.path.join(OUT, "ref_features_cases.json"), "w"))
    json.dump(m, open(os.path.join(OUT, "ref_movegen_cases.json"), "w"))
    print("rules cases", len(r), "checks", sum(len(c["checks"]) for c in r))
    print("features cases", len(f), "bit checks", sum(len(c["bits"]) for c in f))
    print("movegen lists", len(m), "tests", len(set(c["name"].split(":")[0] for c in m)),
          "unparsed", sum(len(c.get("unparsed_expectations", [])) for c in m))
    for c in m:
        for u in c.get("unparsed_expectations", []):
            print("   UNPARSED", c["name"], u)
