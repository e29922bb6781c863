"""Name -> class registry with the reference's call shape: `@MODELS.register_module()` and
`MODELS.build({'NAME': 'VCN_VC', ...})` constructing `cls(cfg)` (models/vcn/utils/registry.py:246-285)."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            if not inspect.isclass(cls):
                raise TypeError(f"module must be a class, but got {type(cls)}")
            names = [cls.__name__] if name is None else ([name] if isinstance(name, str) else list(name))
            for n in names:
                if not force and n in self._module_dict:
                    raise KeyError(f"{n} is already registered in {self._name}")
                self._module_dict[n] = cls
            return cls

        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, **kwargs):
        if not isinstance(cfg, dict):
            raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
        if "NAME" not in cfg:
            raise KeyError(f'`cfg` must contain the key "NAME", but got {cfg}')
        obj_type = cfg.get("NAME")
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(f"{obj_type} is not in the {self._name} registry")
        elif inspect.isclass(obj_type):
            cls = obj_type
        else:
            raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
        return cls(cfg)
