"""Import-only stub of pydantic_settings.BaseSettings: a pydantic BaseModel that fills fields from os.environ."""
import os
from pydantic import BaseModel, ConfigDict


class BaseSettings(BaseModel):
    model_config = ConfigDict(extra="ignore", arbitrary_types_allowed=True)

    def __init__(self, **data):
        for name in type(self).model_fields:
            if name not in data and name in os.environ:
                data[name] = os.environ[name]
        super().__init__(**data)


def SettingsConfigDict(**kw):
    return ConfigDict(extra="ignore", arbitrary_types_allowed=True)
