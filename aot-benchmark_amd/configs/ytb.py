from .default import stage

EngineConfig = stage('ytb')
