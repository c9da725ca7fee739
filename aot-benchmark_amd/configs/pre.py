from .default import stage

EngineConfig = stage('pre')
