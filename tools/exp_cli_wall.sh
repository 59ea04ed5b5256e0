#!/bin/bash
# Measurement helper: the command line on the headline scene, three cold processes in a row: its two timing lines and its --stats line.
cd assets
for i in 1 2 3; do
  timeout -k 10 120 ../raytracer_amd/run -s scenes/meshDragon.json -w 3840 -h 2160 -i /tmp/dragon_4k.png --stats | tail -3 || exit 1
done
