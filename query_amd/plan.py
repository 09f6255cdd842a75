"""Host-side mirror of the reference's plan nodes for the hot path.

Same names and JSON shapes as plan/filter.go:46-53, plan/group.go:54-70,
plan/sequence.go:48-57 and plan/parallel.go:54-67, so a plan built here is
byte-compatible with what the reference planner marshals (EXPLAIN golden:
test/filestore/json/default/cases/case_by_id.json:369-456).  Expressions are
expression.Stringer text.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import List, Optional, Sequence


def field_path(alias: str, *names: str) -> str:
    """Stringer text of a field navigation: field_path("default", "price") -> (`default`.`price`)"""
    s = "`%s`" % alias
    for n in names:
        s = "(%s.`%s`)" % (s, n)
    return s


@dataclass
class Filter:
    condition: str

    def marshal(self) -> dict:
        return {"#operator": "Filter", "condition": self.condition}


@dataclass
class InitialGroup:
    group_keys: List[str] = field(default_factory=list)
    aggregates: List[str] = field(default_factory=list)

    def marshal(self) -> dict:
        # aggregates are de-duplicated and sorted by text, as planner/build_select_sub.go:551-558 does
        return {"#operator": "InitialGroup", "aggregates": list(self.aggregates), "group_keys": list(self.group_keys)}


@dataclass
class IntermediateGroup(InitialGroup):
    def marshal(self) -> dict:
        d = super().marshal()
        d["#operator"] = "IntermediateGroup"
        return d


@dataclass
class FinalGroup(InitialGroup):
    def marshal(self) -> dict:
        d = super().marshal()
        d["#operator"] = "FinalGroup"
        return d


@dataclass
class Sequence:
    children: Sequence[object]

    def marshal(self) -> dict:
        return {"#operator": "Sequence", "~children": [c.marshal() for c in self.children]}


@dataclass
class Parallel:
    child: object
    max_parallelism: int = 0

    def marshal(self) -> dict:
        d = {"#operator": "Parallel", "~child": self.child.marshal()}
        if self.max_parallelism:
            d["maxParallelism"] = self.max_parallelism
        return d


def marshal_json(op) -> str:
    return json.dumps(op.marshal(), sort_keys=True)


def filter_group_plan(condition: Optional[str], group_keys: Optional[Sequence[str]],
                      aggregates: Optional[Sequence[str]], *, filter_only: bool = False) -> str:
    """Plan JSON of Parallel{Sequence[Filter?, InitialGroup?]} as the planner emits it
    (planner/build_select_sub.go:209-211, 276-296)."""
    children: List[object] = []
    if condition:
        children.append(Filter(condition))
    if not filter_only:
        children.append(InitialGroup(list(group_keys or []), list(aggregates or [])))
    return marshal_json(Parallel(Sequence(children)))
