"""Where samples come from: an in-memory table (HF ``datasets.Dataset``, list of dicts) or anything ``datasets.load_dataset``
opens (``/root/reference/ssi/data/sft.py:143-148``, ``cpt.py:89-94``, ``__init__.py:31-56``).  Offline that means local json / csv /
parquet files (``source: json`` + ``data_files``); hub datasets need the network the reference also needs."""

from __future__ import annotations

import logging
from typing import Any, Callable, Sequence

LOGGER = logging.getLogger(__name__)


def load_dataset_subset(source: str, n_samples: int, **load_dataset_kwargs: Any):
    """First ``n_samples`` rows through a streaming read, materialised (``ssi/data/__init__.py:31-56``)."""
    import datasets as hf_datasets
    if "split" not in load_dataset_kwargs:
        raise ValueError("load_dataset_subset requires a 'split' kwarg (e.g. split='train')")
    iterable = hf_datasets.load_dataset(source, streaming=True, **load_dataset_kwargs)
    rows = list(iterable.take(n_samples))
    LOGGER.info(f"Streamed {len(rows)}/{n_samples} samples from {source} (split={load_dataset_kwargs.get('split', '?')})")
    return hf_datasets.Dataset.from_list(rows)


class _Rows:
    """Minimal table over a list of dicts (``len``, integer indexing, ``features``, ``filter``)."""

    def __init__(self, rows: Sequence[dict[str, Any]]):
        self.rows = list(rows)

    def __len__(self) -> int:
        return len(self.rows)

    def __getitem__(self, i: int) -> dict[str, Any]:
        return self.rows[i]

    @property
    def features(self) -> set[str]:
        return set(self.rows[0]) if self.rows else set()

    def filter(self, fn: Callable[[dict[str, Any]], bool]) -> "_Rows":
        return _Rows([r for r in self.rows if fn(r)])


def open_source(source: Any, n_samples: int | None = None, **load_dataset_kwargs: Any):
    if isinstance(source, str):
        import datasets as hf_datasets
        if n_samples is not None:
            return load_dataset_subset(source, n_samples, **load_dataset_kwargs)
        data = hf_datasets.load_dataset(source, **load_dataset_kwargs)
        if not isinstance(data, hf_datasets.Dataset):
            raise TypeError(f"Expected a datasets.Dataset object but found {type(data)}")
        return data
    if isinstance(source, (list, tuple)):
        return _Rows(source[:n_samples] if n_samples is not None else source)
    if hasattr(source, "__len__") and hasattr(source, "__getitem__"):   # datasets.Dataset or any indexable table
        if n_samples is not None and hasattr(source, "select"):
            return source.select(range(min(n_samples, len(source))))
        return source
    raise TypeError(f"cannot open a dataset from {type(source)}")
