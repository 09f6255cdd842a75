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


@dataclass
class Order:
    """plan/order.go:51-79: sort_terms [{expr, desc?}], optional offset / limit expression strings."""
    sort_terms: Sequence[tuple]  # (expression text, descending)
    offset: Optional[int] = None
    limit: Optional[int] = None

    def marshal(self) -> dict:
        terms = []
        for expr, desc in self.sort_terms:
            t = {"expr": expr}
            if desc:
                t["desc"] = True
            terms.append(t)
        d = {"#operator": "Order", "sort_terms": terms}
        if self.offset is not None:
            d["offset"] = str(int(self.offset))
        if self.limit is not None:
            d["limit"] = str(int(self.limit))
        return d


@dataclass
class Limit:
    expr: int

    def marshal(self) -> dict:
        return {"#operator": "Limit", "expr": str(int(self.expr))}


@dataclass
class Offset:
    expr: int

    def marshal(self) -> dict:
        return {"#operator": "Offset", "expr": str(int(self.expr))}


@dataclass
class InitialProject:
    """plan/project.go:73-110: result_terms [{expr, as?}] (star / raw / distinct projections are not built here)."""
    result_terms: Sequence[tuple]  # (expression text, alias or None)

    def marshal(self) -> dict:
        terms = []
        for expr, alias in self.result_terms:
            t = {"expr": expr}
            if alias:
                t["as"] = alias
            terms.append(t)
        return {"#operator": "InitialProject", "result_terms": terms}


@dataclass
class FinalProject:
    def marshal(self) -> dict:
        return {"#operator": "FinalProject"}


def marshal_json(op) -> str:
    return json.dumps(op.marshal(), sort_keys=True)


def filter_group_plan(condition: Optional[str], group_keys: Optional[Sequence[str]],
                      aggregates: Optional[Sequence[str]], *, filter_only: bool = False,
                      order: Optional[Sequence[tuple]] = None, limit: Optional[int] = None,
                      offset: Optional[int] = None, having: Optional[str] = None,
                      project: Optional[Sequence[tuple]] = None) -> str:
    """Plan JSON of Parallel{Sequence[Filter?, InitialGroup?]} as the planner emits it
    (planner/build_select_sub.go:209-211, 276-296).  With `order` / `limit` / `offset` the whole grouped tail
    follows: IntermediateGroup, FinalGroup, Order (which carries offset and limit, plan/order.go:51-79), Offset,
    Limit (planner/build_select.go) — the sort terms must be group keys or aggregates of the plan."""
    children: List[object] = []
    if condition:
        children.append(Filter(condition))
    if not filter_only:
        children.append(InitialGroup(list(group_keys or []), list(aggregates or [])))
    par = Parallel(Sequence(children))
    if order is None and limit is None and offset is None and having is None and project is None:
        return marshal_json(par)
    tail: List[object] = [par, IntermediateGroup(list(group_keys or []), list(aggregates or [])),
                          FinalGroup(list(group_keys or []), list(aggregates or []))]
    if project is not None:
        # HAVING and the projection share the Parallel after FinalGroup (planner/build_select_sub.go:217-235, :295);
        # `project` = [(expression text, alias or None)]
        sub: List[object] = ([Filter(having)] if having is not None else []) + [InitialProject(list(project))]
        tail.append(Parallel(Sequence(sub)))
    elif having is not None:  # HAVING is a Filter over the final groups (planner/build_select_sub.go:295)
        tail.append(Filter(having))
    if order:
        tail.append(Order(list(order), offset, limit))
    else:
        if offset is not None:
            tail.append(Offset(offset))
        if limit is not None:
            tail.append(Limit(limit))
    if project is not None:
        tail.append(FinalProject())
    return marshal_json(Sequence(tail))
